"""Fuzz of the two-stage (float32_prefilter) band against the exact path over random unit latents and poses: python tools/fuzz_prefilter.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
d0, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); d0 = d0.to(dev)
d1, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter"); d1 = d1.to(dev)
B, D, H, W = 8, 40, 64, 64
K = K_for(H, W)
b0 = sdflabel_amd.BatchRenderer(d0, D, K, (W, H), B, device=dev)
b1 = sdflabel_amd.BatchRenderer(d1, D, K, (W, H), B, device=dev)
print("calibrated half-pass deviation %.2e, margin %.4f" % (b1.f16_error, b1.margin))
rng = np.random.default_rng(123)
bad = 0; worst16 = 0.0; tot = 0
for it in range(8):
    lat = rng.standard_normal((B, 3)).astype(np.float32)
    yaw = rng.uniform(-3, 3, B).astype(np.float32)
    tr = np.stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.2, 0.2, B), rng.uniform(2.5, 4.5, B)], 1).astype(np.float32)
    a = [torch.from_numpy(x).to(dev) for x in (yaw, tr, lat)]
    o0 = b0.forward(*a); o1 = b1.forward(*a)
    for b in range(B):
        n0, n1 = int(b0.cnt[b]), int(b1.cnt[b]); tot += 1
        same = n0 == n1 and torch.equal(b0.idx[b, :n0], b1.idx[b, :n1])
        if not same:
            bad += 1; print("band differs: it", it, "crop", b, n0, n1)
        # how close did the half pass come to the margin on rows that matter?  (exact |sdf| of rows the prefilter kept out)
    s0 = b0.sdf.view(B, -1); s1 = b1.sdf.view(B, -1)
    worst16 = max(worst16, float((s0 - s1).abs().max()))
    dm = float((o0["mask"] - o1["mask"]).abs().max()); dc = float((o0["color"] - o1["color"]).abs().max())
    print("iter %d: bands %s, max|mask diff| %.1e max|colour diff| %.1e, candidates/band %.3f" %
          (it, "equal" if bad == 0 else "DIFFER", dm, dc, float(b1.ccnt.sum()) / max(1, float(b1.cnt.sum()))))
print("crops %d, differing bands %d, largest |half - exact| on the grid %.2e (margin %.4f)" % (tot, bad, worst16, b1.margin))
sys.exit(1 if bad else 0)
