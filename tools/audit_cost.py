"""Cost of the two-stage mode's audit: 64 crops x 60 iterations through BatchRefiner (float32_prefilter + candidate reuse, HIP-graph replay) with the
audit's reference values from the split kernel (default), from the exact-f32 kernel, and without audit: python tools/audit_cost.py [--crops 64]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
ap = argparse.ArgumentParser(); ap.add_argument("--crops", type=int, default=64); ap.add_argument("--size", type=int, default=256); a = ap.parse_args()
dev = "cuda"; B = a.crops; H = W = a.size; K = K_for(H, W)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
nocs, lidar = synthetic_targets(dec, 40, K, H, W, dev)
for label, audit, arith in (("audit: split kernel", True, "split"), ("audit: exact f32", True, "float32"), ("no audit", False, "split")):
    d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    d2.prefilter_reuse, d2.prefilter_audit, d2.prefilter_audit_arith = True, audit, arith
    rf = sdflabel_amd.BatchRefiner(d2.to(dev), 40, K, (H, W), B, lidar_cap=4096, device=dev)
    rf.set_crops(crop_params(list(range(B))), nocs.expand(B, 3, H, W), [lidar] * B)
    rf.capture(); rf.optimize(2)
    rf.set_crops(crop_params(list(range(B))), nocs.expand(B, 3, H, W), [lidar] * B)
    torch.cuda.synchronize(); t = time.perf_counter()
    rf.optimize(60); torch.cuda.synchronize(); dt = time.perf_counter() - t
    rep = rf.br.prefilter_report()
    print("%-22s %6.2f crops/s  %6.2f ms per iteration of %d crops  yaw[0] %.6f  hard violations %d  %s" % (label, B / dt, dt / 60 * 1e3, B, float(rf.yaw[0]),
          rep["hard_violations"], rep.get("audit", "")), flush=True)
    del rf
