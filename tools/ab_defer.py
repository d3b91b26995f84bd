"""A/B inside one process: the drop-in crop-iteration with / without queueing the band kernels ahead of the host read of N (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sdflabel_amd
from sdflabel_amd import grid as grid_mod
from sdflabel_amd.fixtures import ASSET
dev = torch.device("cuda", 0)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(bench.D, dev)
renderer = sdflabel_amd.Rasterer(torch.from_numpy(bench.K_for(bench.H, bench.W)), (bench.W, bench.H)).to(dev)
crop = bench.Crop(0, dev)
for _ in range(20): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(6):
    for mode in (True, False):
        grid_mod._DEFER = mode
        for _ in range(5): bench.crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): bench.crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t) / 100 * 1e3)
for mode in (True, False):
    print("defer" if mode else "no-defer", " ".join("%.3f" % v for v in res[mode]), "min %.3f" % min(res[mode]))
