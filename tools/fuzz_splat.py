"""One-off fuzz of the splat kernels against the oracle at candidate densities beyond the unit tests (multi-round and overflow paths of the
four-wave forward): python tools/fuzz_splat.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import traceback
from tests import test_gpu_parity as t
bad = 0
for H, W, n in ((24, 24, 600), (24, 24, 1500), (16, 40, 1200), (33, 17, 900), (64, 64, 3000), (8, 8, 1100), (9, 71, 700)):
    for kshift in ((0.0, 0.0), (-23.5, 6.25), (41.0, -17.5)):          # centred and off-centre principal points (cropped intrinsics)
        try:
            t.test_splat_forward_backward_vs_oracle(H, W, n, kshift)
            print("ok", H, W, n, kshift, flush=True)
        except Exception as e:
            bad += 1
            print("FAIL", H, W, n, kshift, repr(e)[:300], flush=True)
for prim, bg in (("circle", True), ("circle_opt", True), ("disc", True), ("circle", False)):
    try:
        t.test_secondary_primitives_and_bg_golden(prim, bg)
        print("ok", prim, bg)
    except Exception as e:
        bad += 1; print("FAIL", prim, bg, repr(e)[:300])
sys.exit(1 if bad else 0)
