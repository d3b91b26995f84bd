import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets
dev = "cuda"
mode, seq = sys.argv[1], sys.argv[2]
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); d16 = d16.to(dev)
H, W = 200, 300
K = K_for(H, W)
shapes = {"same": ((200, 300), (200, 300)), "grow": ((200, 300), (148, 442)), "shrink": ((200, 300), (100, 150)), "tall": ((200, 300), (300, 200)),
          "wide": ((200, 300), (199, 301))}[seq]
rf = sdflabel_amd.BatchRefiner(d16, 40, K, (H, W), 1, lidar_cap=1024, device=dev, render="trace", max_pixels=65536, max_side=1024, tracer_kwargs=kw)
for hw in shapes:
    Kc = K_for(*hw)
    n1, l1 = synthetic_targets(dec, 40, Kc, hw[0], hw[1], dev)
    rf.set_crops(crop_params([0]), [n1[0]], [l1[:1024]], K=Kc, crop_sizes=[hw])
    if rf._replay is None and mode == "graph":
        rf.capture()
    rf.optimize(5); torch.cuda.synchronize()
    print(mode, seq, hw, rf.results()[0][0, :4].tolist(), rf.tr.stats(), flush=True)
print("done", mode, seq, kw)
