"""Host-side profile of the drop-in crop-iteration (development aid)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev = torch.device("cuda", 0)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(bench.D, dev)
renderer = sdflabel_amd.Rasterer(torch.from_numpy(bench.K_for(bench.H, bench.W)), (bench.W, bench.H)).to(dev)
crop = bench.Crop(0, dev)
for _ in range(5): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize()
import time
t=time.perf_counter()
for _ in range(20): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter()-t)/20*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(48); print(s.getvalue()[:9000])
