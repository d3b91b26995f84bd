import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
dev = "cuda"
which = sys.argv[1]
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); d16 = d16.to(dev)
if which == "sharded":
    H = W = 256; B = 8
    K = K_for(H, W)
    rf = sdflabel_amd.BatchRefiner(d16, 40, K, (H, W), B, lidar_cap=4096, device=dev, render="trace")
    nocs1, lidar = synthetic_targets(dec, 40, K, H, W, dev)
    rf.set_crops(crop_params(list(range(B))), nocs1.expand(B, 3, H, W), [lidar] * B)
    rf.capture(); rf.optimize(5); torch.cuda.synchronize(); print("sharded ok", rf.results()[0][0])
else:
    cap = int(sys.argv[2]); side = int(sys.argv[3]); H, W = int(sys.argv[4]), int(sys.argv[5])
    K = K_for(H, W)
    tr = sdflabel_amd.SphereTracer(d16, K, (W, H), 1, device=dev, max_pixels=cap, max_side=side, points=True)
    a = [torch.tensor([0.7], device=dev), torch.tensor([[0.0, 0.0, 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev)]
    tr.render(*a); torch.cuda.synchronize(); print("render ok", tr.stats())
    tr.backward(g_color=torch.ones_like(tr.color), g_xyzf=torch.ones_like(tr.xyzf), surfel=True); torch.cuda.synchronize(); print("bwd ok")
