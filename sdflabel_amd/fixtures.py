"""Synthetic fixtures shared by bench.py, tools/ and the tests: the committed decoder assets and the synthetic refinement problems of
SURVEY.md §8(d).  No pretrained DeepSDF / CSS weights and no KITTI data exist offline, so the workloads are built from a decoder fitted to
an analytic shape (tools/fit_decoder.py) and targets rendered from a ground-truth pose (the a-harness of SURVEY.md §8).

Nothing here is on the product path of a caller that brings its own decoder and crops.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ASSET_DIR = os.path.join(_HERE, "assets")
ASSET = os.path.join(ASSET_DIR, "deepsdf_synth")                       # rounded-box fit, weight-norm 8x512, L = 3 (the bench decoder)
ASSET_ELLIPSOID = os.path.join(ASSET_DIR, "deepsdf_synth_ellipsoid")   # second fixture: ellipsoid fit (r03)
ASSET_ELLIPSOID_LN = os.path.join(ASSET_DIR, "deepsdf_synth_ellipsoid_ln")   # ... and its LayerNorm variant (weight_norm=False)

GT_YAW, GT_TRANS, GT_LATENT, GT_SCALE = 0.6, (0.0, 0.0, 3.5), (0.3, -0.5, 0.8), 2.0


def fitted_state(asset=ASSET):
    """A committed decoder as {key: float32 ndarray} + its NetworkSpecs (deepsdf/workspace.py:167-180 on-disk format)."""
    import torch
    st = torch.load(asset + ".pt", map_location="cpu")["model_state_dict"]
    st = {k[len("module."):] if k.startswith("module.") else k: v.float().numpy() for k, v in st.items()}
    spec = json.load(open(asset + ".json"))["NetworkSpecs"]
    return st, spec


def K_for(H, W):
    """centred synthetic intrinsics of SURVEY.md §8(d): f = 45 H / 32, principal point at the crop centre"""
    f = 45.0 * H / 32.0
    return np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)


def crop_start(index):
    """Initial parameters of synthetic crop `index`: the ground truth (yaw .6, t (0, 0, 3.5), latent (.3, -.5, .8), scale 2) perturbed by a
    jitter seeded with the crop index (the same on every rank).  Returns (yaw(1,), trans(3,), latent(3,)) float32 arrays."""
    import torch
    jit = torch.rand(7, generator=torch.Generator().manual_seed(1 + int(index))).numpy().astype(np.float32)
    yaw = np.float32(GT_YAW) + np.float32(0.1) + np.float32(0.1) * jit[0:1]
    trans = np.asarray(GT_TRANS, np.float32) + np.asarray([0.1, 0.05, -0.3], np.float32) * jit[1:4]
    latent = np.asarray(GT_LATENT, np.float32) + np.float32(0.2) * (jit[4:7] - np.float32(0.5))
    return yaw.astype(np.float32), trans.astype(np.float32), latent.astype(np.float32)


def crop_params(indices):
    """{'yaw' (n,), 'trans' (n,3), 'scale' (n,), 'latent' (n,3)} float32 arrays for the synthetic crops `indices`"""
    st = [crop_start(i) for i in indices]
    n = len(st)
    return {"yaw": np.concatenate([s[0] for s in st]) if n else np.zeros((0,), np.float32),
            "trans": np.stack([s[1] for s in st]) if n else np.zeros((0, 3), np.float32),
            "scale": np.full((n,), GT_SCALE, np.float32),
            "latent": np.stack([s[2] for s in st]) if n else np.zeros((0, 3), np.float32)}


def synthetic_targets(decoder, density, K, H, W, device, lidar_stride=2):
    """Target NOCS image (1,3,H,W) and lidar-like cloud (M,3) of the ground-truth pose, rendered with the exact-f32 path of `decoder`'s
    weights (what refine_css_demo.py:107-131 would supply from the CSS net and the lidar sweep).  GPU only."""
    import torch
    from .batch import BatchRenderer
    gt = BatchRenderer(decoder, density, K, (W, H), 1, device=device)
    o = gt.forward(torch.tensor([GT_YAW], device=device), torch.tensor([GT_TRANS], device=device), torch.tensor([GT_LATENT], device=device))
    nf = int(o["nf"][0])
    lidar = (o["xyzf"][0, :nf] * GT_SCALE)[::lidar_stride].cpu().numpy()
    return o["color"].clone(), lidar
