"""Splat forward / backward at the reference's shipped rendering_area (32): tile candidate statistics and launch times per batch size.
    python tools/splat32_diag.py [--area 32] [B ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, GT_LATENT, GT_YAW, kitti_like_crops

ap = argparse.ArgumentParser()
ap.add_argument("--area", type=int, default=32)
ap.add_argument("B", nargs="*", type=int, default=[1, 4, 16])
args = ap.parse_args()
dev = torch.device("cuda", 0)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
dec = dec.to(dev)
for B in args.B:
    shapes, Ks, gts = kitti_like_crops(args.area, B)
    pmax = max(1024, 1 << (max(h * w for h, w in shapes) - 1).bit_length())
    br = sdflabel_amd.BatchRenderer(dec, 40, np.stack(Ks), (shapes[0][1], shapes[0][0]), B, device=dev, max_pixels=pmax)
    br.set_extents([(w, h) for h, w in shapes], np.stack(Ks))
    br.set_params(torch.full((B,), GT_YAW, device=dev), torch.from_numpy(np.stack(gts)).to(dev), torch.tensor([list(GT_LATENT)] * B, device=dev))
    g3 = torch.ones_like(br.color)
    gx = torch.ones_like(br.xyzf)
    ev = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for _ in range(3):
        br.forward(); br.backward(g_color=g3, g_xyzf=gx)
    ts = {"splat_fwd": [], "splat_bwd": [], "jacobian": []}
    for _ in range(10):
        e = {k: ev() for k in ts}
        br.forward(events=e); br.backward(g_color=g3, g_xyzf=gx, events=e)
        torch.cuda.synchronize()
        for k in ts:
            ts[k].append(e[k][0].elapsed_time(e[k][1]) * 1e3)
    boxes = br.boxes.cpu().numpy()
    cnt = br.cnt.cpu().numpy()
    per_tile, sizes = [], []
    for b in range(B):
        h, w = shapes[b]
        bb = boxes[b, :cnt[b]]
        ok = (bb[:, 0] <= bb[:, 2]) & (bb[:, 1] <= bb[:, 3])
        bb = bb[ok]
        sizes.append(((bb[:, 2] - bb[:, 0] + 1) * (bb[:, 3] - bb[:, 1] + 1)))
        for ty in range((h + 7) // 8):
            for tx in range((w + 7) // 8):
                X0, Y0 = tx * 8, ty * 8
                X1, Y1 = min(X0 + 7, w - 1), min(Y0 + 7, h - 1)
                per_tile.append(int((~((bb[:, 0] > X1) | (bb[:, 2] < X0) | (bb[:, 1] > Y1) | (bb[:, 3] < Y0))).sum()))
    per_tile, sizes = np.asarray(per_tile), np.concatenate(sizes)
    print("area %d B=%d binned=%s: surfels %d front-facing %d boxes on screen %d (box pixels mean %.1f max %d) | tiles %d: candidates per tile mean %.0f "
          "median %.0f max %d | splat fwd %.1f us  bwd %.1f us  jacobian %.1f us"
          % (args.area, B, br.binned, int(cnt.sum()), int(br.fcnt.sum()), len(sizes), sizes.mean(), sizes.max(), len(per_tile), per_tile.mean(), np.median(per_tile),
             per_tile.max(), min(ts["splat_fwd"]), min(ts["splat_bwd"]), min(ts["jacobian"])), flush=True)
    del br
