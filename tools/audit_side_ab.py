import sys, time, torch
sys.path.insert(0, "/root/repo")
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
dev = "cuda"
d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); d32 = d32.to(dev)
K = K_for(256, 256)
nocs1, lidar = synthetic_targets(d32, 40, K, 256, 256, dev)
for prec in (torch.float16, torch.float32):
    for side in (16, 64):
        d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec); d.candidate_reuse = True; d.candidate_audit_side_max_crops = side; d = d.to(dev)
        rf = sdflabel_amd.BatchRefiner(d, 40, K, (256, 256), 64, lidar_cap=4096, device=dev)
        p = crop_params(list(range(64)))
        rf.set_crops(p, nocs1.expand(64, 3, 256, 256), [lidar] * 64)
        rf.capture(); rf.optimize(2)
        best = 1e9
        for rep in range(3):
            p = crop_params(list(range(64 * rep, 64 * rep + 64)))
            rf.set_crops(p, nocs1.expand(64, 3, 256, 256), [lidar] * 64)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rf.optimize(60); rf.results()
            best = min(best, time.perf_counter() - t0)
        print(prec, "audit side stream up to", side, "crops: ms per chunk %.1f -> %.1f crops/s" % (best * 1e3, 64 / best), flush=True)
        del rf
