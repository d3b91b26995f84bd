"""CPU time of the library's profiler ranges (sdfr::...) and of the autograd nodes in one drop-in crop-iteration (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench, sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev = torch.device("cuda", 0)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(bench.D, dev)
renderer = sdflabel_amd.Rasterer(torch.from_numpy(bench.K_for(bench.H, bench.W)), (bench.W, bench.H)).to(dev)
crop = bench.Crop(0, dev)
for _ in range(5): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize()
N = 20
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N): bench.crop_iteration(dec, grid, renderer, crop)
    torch.cuda.synchronize()
rows = [(e.key, e.count / N, e.cpu_time_total / N, e.self_cpu_time_total / N) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[2])
print("%-70s %6s %10s %10s" % ("name", "calls", "cpu us", "self us"))
for r in rows[:45]:
    print("%-70s %6.1f %10.1f %10.1f" % (r[0][:70], r[1], r[2], r[3]))
