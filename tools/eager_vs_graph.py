import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
H = W = 256; D = 40
for prec in (torch.float16, "float32_split", torch.float32):
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec); dec = dec.to(dev)
    br = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=dev)
    br.set_params(torch.tensor([0.7], device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
    ones3 = torch.ones(1, 3, H, W, device=dev); ones1 = torch.ones(1, 1, H, W, device=dev); onesx = torch.ones(1, br.cap, 3, device=dev)
    def step():
        br.forward(); br.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): step()
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize(); te = time.perf_counter() - t
    replay = br.capture(lambda o: dict(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx))
    for _ in range(5): replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): replay()
    torch.cuda.synchronize(); tg = time.perf_counter() - t
    print(prec, "eager %.3f ms/step (CPU issue %.3f)  graph %.3f ms/step" % (te / 200 * 1e3, t_issue / 200 * 1e3, tg / 200 * 1e3))
