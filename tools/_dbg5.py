import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
dev = "cuda"
var = sys.argv[1]; ragged = sys.argv[2] == "ragged"; render = sys.argv[3]
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); d16 = d16.to(dev)
H, W = 200, 300
K = K_for(H, W)
n1, l1 = synthetic_targets(dec, 40, K, H, W, dev)
kw = dict(max_pixels=65536, max_side=1024) if ragged else {}
rf = sdflabel_amd.BatchRefiner(d16, 40, K, (H, W), 1, lidar_cap=1024, device=dev, render=render, **kw)
def setc():
    if ragged: rf.set_crops(crop_params([0]), [n1[0]], [l1[:1024]], K=K, crop_sizes=[(H, W)])
    else: rf.set_crops(crop_params([0]), n1, [l1[:1024]])
setc()
rf.capture(); rf.optimize(5); torch.cuda.synchronize(); print("first ok", flush=True)
if var == "setcrops": setc()
elif var == "targets": synthetic_targets(dec, 40, K, H, W, dev)
elif var == "alloc": x = [torch.zeros(1 << 20, device=dev) for _ in range(50)]; del x
elif var == "results": print(rf.results()[0][0, :4].tolist())
elif var == "stats": print(rf.tr.stats())
rf.optimize(5); torch.cuda.synchronize(); print("second ok", var, sys.argv[2], render, flush=True)
