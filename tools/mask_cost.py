"""What the 32 MB of ReLU masks per crop cost the exact-f32 grid forward: forward with and without mask saving at 1, 8 and
64 crops per launch, and the two band-Jacobian variants that go with them (mask-fed backward-only vs recomputing) on the real band rows.
python tools/mask_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sdflabel_amd
from sdflabel_amd import _lib
from sdflabel_amd.fixtures import ASSET, K_for, crop_params

dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
dec = dec.to(dev)
L = _lib.lib()
P, st = _lib.ptr, None


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for B in (1, 8, 64):
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(256, 256), (256, 256), B, device=dev)
    p = {k: torch.from_numpy(v).to(dev) for k, v in crop_params(list(range(B))).items()}
    br.forward(p["yaw"], p["trans"], p["latent"])
    torch.cuda.synchronize()
    G, cap = br.G, br.cap
    s = _lib.stream_ptr()
    f_mask = timed(lambda: L.sdfr_mlp_forward(br.handle.h, P(br.inputs), B * G, P(br.sdf), P(br.mask_ws), s))
    f_nomask = timed(lambda: L.sdfr_mlp_forward(br.handle.h, P(br.inputs), B * G, P(br.sdf), None, s))
    j_fed = timed(lambda: L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), G, B, P(br.idx), cap, P(br.cnt), P(br.J), P(br.sdf_band), P(br.sdf), P(br.mask_ws), 0, s))
    j_rec = timed(lambda: L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), G, B, P(br.idx), cap, P(br.cnt), P(br.J), P(br.sdf_band), None, None, 0, s))
    h_mask = timed(lambda: L.sdfr_mlp_forward_f16(br.handle.h, P(br.inputs), B * G, P(br.sdf), P(br.mask_ws), s))
    h_nomask = timed(lambda: L.sdfr_mlp_forward_f16(br.handle.h, P(br.inputs), B * G, P(br.sdf), None, s))
    print("B=%2d float16 forward: with masks %.4f ms (%.4f per crop), without %.4f ms (%.4f per crop): masks cost %.1f %%"
          % (B, h_mask, h_mask / B, h_nomask, h_nomask / B, 100 * (h_mask - h_nomask) / h_mask), flush=True)
    L.sdfr_mlp_forward(br.handle.h, P(br.inputs), B * G, P(br.sdf), P(br.mask_ws), s)         # restore the f32 masks for the Jacobian timings below
    rows = int(br.cnt.sum())
    print("B=%2d (%6d band rows): forward with masks %.3f ms, without %.3f ms (%.1f %%); Jacobian mask-fed %.3f ms, recomputing %.3f ms; "
          "pair with masks %.3f ms vs mask-free pair %.3f ms  [per crop: %.3f vs %.3f]"
          % (B, rows, f_mask, f_nomask, 100 * (f_mask - f_nomask) / f_mask, j_fed, j_rec, f_mask + j_fed, f_nomask + j_rec,
             (f_mask + j_fed) / B, (f_nomask + j_rec) / B), flush=True)
    del br
