"""A/B: one 64-crop refiner against TWO refiners replayed on two streams (the decoder passes of one chunk beside the splat / loss kernels of the
other): python tools/two_stream_ab.py [float16|float32] [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
dev = "cuda"
prec = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "float32") else torch.float16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); d32 = d32.to(dev)
K = K_for(256, 256)
nocs1, lidar = synthetic_targets(d32, 40, K, 256, 256, dev)
d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec); d.candidate_reuse = True; d = d.to(dev)


def make(off):
    rf = sdflabel_amd.BatchRefiner(d, 40, K, (256, 256), B, lidar_cap=4096, device=dev)
    rf.set_crops(crop_params(list(range(off, off + B))), nocs1.expand(B, 3, 256, 256), [lidar] * B)
    rf.capture(); rf.optimize(2)
    return rf


def reset(rf, off):
    rf.set_crops(crop_params(list(range(off, off + B))), nocs1.expand(B, 3, 256, 256), [lidar] * B)


a = make(0)
best = 1e9
for rep in range(3):
    reset(a, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a.optimize(60); ra = a.results()[0].clone()
    reset(a, B)
    a.optimize(60); rb = a.results()[0].clone()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("one refiner, two chunks of %d in turn: %.1f ms -> %.1f crops/s" % (B, best * 1e3, 2 * B / best), flush=True)
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rfs = [a] + [make(B * k) for k in range(1, NS)]
sts = [torch.cuda.Stream() for _ in range(NS)]
best = 1e9
for rep in range(3):
    reset(a, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for k in range(NS):
        if k:
            reset(a, B * k)
        a.optimize(60); outs.append(a.results()[0].clone())
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("one refiner, %d chunks of %d in turn: %.1f ms -> %.1f crops/s" % (NS, B, best * 1e3, NS * B / best), flush=True)
best2 = 1e9
for rep in range(3):
    for k, rf in enumerate(rfs):
        reset(rf, B * k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for st in sts:
        st.wait_stream(torch.cuda.current_stream())
    for it in range(60):
        for rf, st in zip(rfs, sts):
            with torch.cuda.stream(st):
                rf.optimize(1)
    for st in sts:
        torch.cuda.current_stream().wait_stream(st)
    outs2 = [rf.results()[0].clone() for rf in rfs]
    torch.cuda.synchronize()
    best2 = min(best2, time.perf_counter() - t0)
print("%d refiners on %d streams: %.1f ms -> %.1f crops/s (x%.3f); same results: %s" % (NS, NS, best2 * 1e3, NS * B / best2, best / best2,
      all(torch.equal(x, y) for x, y in zip(outs, outs2))), flush=True)
