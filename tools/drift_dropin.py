"""ms per drop-in crop-iteration, event-timed decoder kernel and allocated device memory per block of 100 iterations (development aid: a leak
or a slow drift shows up here, not in a 20-step timing)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev = torch.device("cuda", 0)
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(bench.D, dev)
renderer = sdflabel_amd.Rasterer(torch.from_numpy(bench.K_for(bench.H, bench.W)), (bench.W, bench.H)).to(dev)
crop = bench.Crop(0, dev)
for _ in range(10): bench.crop_iteration(dec, grid, renderer, crop)
torch.cuda.synchronize()
for blk in range(10):
    t = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    for i in range(100): bench.crop_iteration(dec, grid, renderer, crop, ev if i == 99 else None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 100 * 1e3
    print("block %d: %.3f ms/iter, decoder kernel %.3f ms, reserved %.0f MB, allocated %.0f MB, gc %s" % (blk, dt, ev[0].elapsed_time(ev[1]), torch.cuda.memory_reserved() / 1e6, torch.cuda.memory_allocated() / 1e6, gc.get_count()), flush=True)
