"""Fit the synthetic 8x512 DeepSDF decoder fixture (runs in the build container only).

No pretrained DeepSDF weights exist in this environment, and a
random-init decoder puts every grid point inside the |sdf|<0.03 band.  This
script instantiates the REFERENCE Decoder class
(sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py:9-75, imported from
/root/reference, never copied) with the standard DeepSDF spec
(dims 8x512, latent_in=[4], norm_layers=0..7, weight_norm=True, L=3) and fits it
(seed 1) to an analytic latent-conditioned rounded-box SDF.  The result is saved
in the reference's own on-disk format (<name>.json specs + <name>.pt state with
DataParallel 'module.' prefix, workspace.py:167-180) with fp16 tensors to keep
the fixture small (3.7 MB).

r03: `--shape ellipsoid` fits a second fixture (a latent-conditioned ellipsoid: N ~ 0.7 k band surfels at D = 40 instead of the
box's 2.7 k, smooth normals instead of flat faces) and `--layer-norm` the reference's LayerNorm variant of the decoder
(weight_norm=False, deep_sdf_decoder_scale.py:56-57,99-101), so that full-size parity is not pinned on a single set of weights.

usage: python tools/fit_decoder.py [--steps 1500] [--shape box|ellipsoid] [--layer-norm] [--out sdflabel_amd/assets/deepsdf_synth]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import  # noqa: E402

_ref_import.setup()

import numpy as np  # noqa: E402
import torch  # noqa: E402

SPECS = {
    "Description": "synthetic latent-conditioned rounded box, fitted by tools/fit_decoder.py (seed 1)",
    "NetworkArch": "deep_sdf_decoder_scale",
    "CodeLength": 3,
    "NetworkSpecs": {
        "dims": [512] * 8,
        "dropout": [0, 1, 2, 3, 4, 5, 6, 7],
        "dropout_prob": 0.2,
        "norm_layers": [0, 1, 2, 3, 4, 5, 6, 7],
        "latent_in": [4],
        "xyz_in_all": False,
        "use_tanh": False,
        "latent_dropout": False,
        "weight_norm": True,
    },
}


def half_extents(lat):
    """b(latent): car-like box, latent on the unit sphere (optimizer.py:96 normalises it)."""
    base = lat.new_tensor([0.45, 0.35, 0.90])
    amp = lat.new_tensor([0.08, 0.06, 0.07])
    return base + amp * lat


def sd_round_box(x, b, r=0.05):
    q = x.abs() - (b - r)
    outside = torch.clamp(q, min=0).norm(dim=-1)
    inside = torch.clamp(q.max(dim=-1).values, max=0)
    return outside + inside - r


def radii(lat):
    """r(latent): car-sized ellipsoid"""
    base = lat.new_tensor([0.50, 0.36, 0.82])
    amp = lat.new_tensor([0.07, 0.05, 0.08])
    return base + amp * lat


def sd_ellipsoid(x, r):
    """first-order distance estimate k0 (k0 - 1) / k1 (exact on the surface, Lipschitz-like elsewhere): all the fixture needs is a
    clean zero level set with a well-behaved gradient around it"""
    k0 = (x / r).norm(dim=-1)
    k1 = (x / (r * r)).norm(dim=-1)
    return k0 * (k0 - 1.0) / torch.clamp(k1, min=1e-6)


SHAPE = "box"


def sample_batch(n, gen):
    lat = torch.randn(n // 64, 3, generator=gen)
    lat = lat / lat.norm(dim=1, keepdim=True)
    lat = lat.repeat_interleave(64, 0)
    if SHAPE == "ellipsoid":
        r = radii(lat)
        xu = (torch.rand(n, 3, generator=gen) * 2 - 1) * 1.02
        d = torch.randn(n, 3, generator=gen)
        d = d / d.norm(dim=1, keepdim=True)
        xs = d * r + torch.randn(n, 3, generator=gen) * 0.03          # near the surface
        pick = torch.rand(n, generator=gen) < 0.5
        x = torch.where(pick[:, None], xu, xs)
        return torch.cat([lat, x], 1), sd_ellipsoid(x, r)
    b = half_extents(lat)
    # half uniform in the cube, half near the box surface
    xu = (torch.rand(n, 3, generator=gen) * 2 - 1) * 1.02
    # near-surface: random point in the box, snap one random axis to +-b, then jitter
    xs = (torch.rand(n, 3, generator=gen) * 2 - 1) * b
    ax = torch.randint(0, 3, (n,), generator=gen)
    sgn = (torch.randint(0, 2, (n,), generator=gen) * 2 - 1).float()
    xs[torch.arange(n), ax] = sgn * b[torch.arange(n), ax]
    xs = xs + torch.randn(n, 3, generator=gen) * 0.03
    pick = torch.rand(n, generator=gen) < 0.5
    x = torch.where(pick[:, None], xu, xs)
    sd = sd_round_box(x, b)
    return torch.cat([lat, x], 1), sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "sdflabel_amd", "assets", "deepsdf_synth"))
    ap.add_argument("--shape", default="box", choices=("box", "ellipsoid"))
    ap.add_argument("--layer-norm", action="store_true", help="weight_norm=False: the reference's LayerNorm variant")
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    global SHAPE
    SHAPE = args.shape
    if args.threads:
        torch.set_num_threads(args.threads)

    torch.manual_seed(1)
    np.random.seed(1)
    gen = torch.Generator().manual_seed(1)
    from deepsdf.networks.deep_sdf_decoder_scale import Decoder  # the reference class

    spec = dict(SPECS["NetworkSpecs"])
    if args.layer_norm:
        spec["weight_norm"] = False
    specs = dict(SPECS, NetworkSpecs=spec)
    if args.shape != "box" or args.layer_norm:
        specs["Description"] = "synthetic latent-conditioned %s%s, fitted by tools/fit_decoder.py (seed 1)" % (
            "rounded box" if args.shape == "box" else "ellipsoid", ", LayerNorm variant" if args.layer_norm else "")
    dec = Decoder(SPECS["CodeLength"], **spec)
    dec.eval()  # dropout off: the renderer always evaluates in eval mode (workspace.py:185-186)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(args.steps // 3, 1), gamma=0.4)
    t0 = time.time()
    for it in range(args.steps):
        inp, sd = sample_batch(args.batch, gen)
        pred, _ = dec(inp)
        pred = pred[:, 0]
        tgt = torch.tanh(sd)
        near = (torch.clamp(pred, -0.1, 0.1) - torch.clamp(tgt, -0.1, 0.1)).abs().mean()
        far = (pred - tgt).abs().mean()
        loss = near + 0.2 * far
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if it % 50 == 0 or it == args.steps - 1:
            print(f"it {it} loss {loss.item():.5f} near {near.item():.5f} far {far.item():.5f} t {time.time()-t0:.0f}s", flush=True)

    out = os.path.abspath(args.out)
    state = {"module." + k: v.detach().to(torch.float16) for k, v in dec.state_dict().items()}
    torch.save({"epoch": args.steps, "model_state_dict": state}, out + ".pt")
    with open(out + ".json", "w") as f:
        json.dump(specs, f, indent=1)
    print("saved", out + ".pt")


if __name__ == "__main__":
    main()
