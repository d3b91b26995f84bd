import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
prec = sys.argv[1]; kw = eval(sys.argv[2]); what = sys.argv[3] if len(sys.argv) > 3 else "render"
d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16 if prec == "f16" else torch.float32); d = d.to(dev)
H, W = 200, 300
tr = sdflabel_amd.SphereTracer(d, K_for(H, W), (W, H), 1, device=dev, **kw)
a = [torch.tensor([0.7], device=dev), torch.tensor([[0.0, 0.0, 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev)]
tr.render(*a); torch.cuda.synchronize()
gc = torch.ones_like(tr.color)
def it():
    tr.render()
    if what == "bwd": tr.backward(g_color=gc)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): it()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): it()
for _ in range(3): g.replay()
torch.cuda.synchronize(); print("first ok", flush=True)
x = tr.counters.cpu(); y = torch.zeros(1000, device=dev) + 1
for _ in range(3): g.replay()
torch.cuda.synchronize(); print("second ok", prec, kw, what, tr.stats(), flush=True)
