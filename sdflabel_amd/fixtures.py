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


def kitti_like_crops(area, n=32, seed=11):
    """n crops as the reference PIPELINE produces them (utils/refinement.py:586-609 adjust_intrinsics_crop): every annotation its own crop size
    (area-normalised to rendering_area^2 = `area`^2, aspect of a car's 2-D box kept) and its own intrinsics (principal point moved by the box
    corner, so it usually lies far outside the crop).  Returns (shapes [(H, W)], Ks [3x3 float32], gts [trans(3,) of the ground-truth pose that
    centres the object in the crop]).  Shared by bench.py, tools/ and the tests."""
    rng = np.random.default_rng(seed)
    boxes_w = rng.uniform(60, 420, n)
    boxes_h = boxes_w / rng.uniform(1.2, 3.2, n)                                         # cars: 1.2 ... 3.2 times as wide as high
    shapes, Ks, gts = [], [], []
    for bw, bh in zip(boxes_w, boxes_h):
        r = np.sqrt(area * area / (bh * bw))
        Hc, Wc = int(bh * r), int(bw * r)                                                 # crop_size.int() (:603)
        f = 1.15 * Hc * 3.5 / 2.0                                                         # the object (a 2-unit cube at z = 3.5) about fills the crop's height
        cx, cy = rng.uniform(-1.0 * Wc, 2.0 * Wc), rng.uniform(0.2 * Hc, 0.8 * Hc)        # principal point far outside the crop, as after the box-corner shift
        shapes.append((Hc, Wc))
        Ks.append(np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], np.float32))
        gts.append(np.array([3.5 * (Wc / 2.0 - cx) / f, 3.5 * (Hc / 2.0 - cy) / f, 3.5], np.float32))
    return shapes, Ks, gts


def kitti_like_problems(decoder32, density, area, n, device, seed=11):
    """The refinement problems of `kitti_like_crops`: per crop the target NOCS image (3, H_b, W_b) (CPU tensor, rendered from the ground-truth pose
    with the exact-f32 decoder in ONE ragged batch), the lidar-like cloud (M_b, 3) and perturbed initial parameters {'yaw', 'trans', 'scale',
    'latent'} (numpy, as pipelines/refine_css.py:186-190 builds them).  Returns (shapes, Ks, targets, lidars, starts).  GPU only."""
    import torch
    from .batch import BatchRenderer
    shapes, Ks, gts = kitti_like_crops(area, n, seed)
    pmax = 1 << (max(h * w for h, w in shapes) - 1).bit_length()
    gtr = BatchRenderer(decoder32, density, np.stack(Ks), (shapes[0][1], shapes[0][0]), n, device=device, max_pixels=pmax)
    gtr.set_extents([(w, h) for h, w in shapes], np.stack(Ks))
    o = gtr.forward(torch.full((n,), GT_YAW, device=device), torch.from_numpy(np.stack(gts)).to(device),
                    torch.tensor([list(GT_LATENT)] * n, device=device))
    nfs = o["nf"].tolist()
    targets = [gtr.image(b, "color").clone().cpu() for b in range(n)]
    lidars = [(o["xyzf"][b, :nfs[b]] * GT_SCALE)[::2].cpu().numpy() for b in range(n)]
    starts = []
    for b in range(n):
        y0, t0_, l0 = crop_start(b)
        starts.append({"yaw": y0.copy(), "trans": (gts[b] + (t0_ - np.asarray(GT_TRANS, np.float32))).astype(np.float32),
                       "scale": np.array([GT_SCALE], np.float32), "latent": l0.copy()})
    return shapes, Ks, targets, lidars, starts
