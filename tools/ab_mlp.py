"""A/B the decoder-forward kernel variants (SDFR_MLP_VARIANT), interleaved rounds in separate processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os; sys.path.insert(0, %r)
import torch, torch.nn.functional as F, sdflabel_amd
from tests._util import ASSET
dev="cuda"; dec,_=sdflabel_amd.setup_dsdf(ASSET+".pt"); dec=dec.to(dev)
grid=sdflabel_amd.Grid3D(40,dev); lat=F.normalize(torch.tensor([0.3,-0.5,0.8],device=dev),dim=0)
inp=torch.cat([lat.expand(grid.points.size(0),-1),grid.points],1).contiguous()
with torch.no_grad():
    for _ in range(3): out=dec(inp)[0]
    torch.cuda.synchronize()
    ts=[]
    for r in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out=dec(inp)[0]
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
print("%%.4f %%.4f %%.10f" %% (min(ts), sorted(ts)[2], float(out.double().sum())))
''' % ROOT
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, SDFR_MLP_VARIANT=str(v))
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print("round", rnd, "variant", v, "min/median ms, checksum:", out.stdout.strip() or out.stderr[-300:], flush=True)
