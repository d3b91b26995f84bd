"""A/B decoder-forward variants through the C ABI without mask saving (SDFR_MLP_VARIANT / SDFR_F16_VARIANT), separate processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os; sys.path.insert(0, %r)
import torch, torch.nn.functional as F, sdflabel_amd
from sdflabel_amd import _lib
from tests._util import ASSET
dev="cuda"; dec,_=sdflabel_amd.setup_dsdf(ASSET+".pt"); dec=dec.to(dev)
h = dec.handle(torch.device(dev,0)).h
grid=sdflabel_amd.Grid3D(40,dev); lat=F.normalize(torch.tensor([0.3,-0.5,0.8],device=dev),dim=0)
B = int(sys.argv[2])
inp=torch.cat([lat.expand(grid.points.size(0),-1),grid.points],1).repeat(B,1).contiguous()
out=torch.empty(inp.shape[0],device=dev)
L=_lib.lib(); fn = L.sdfr_mlp_forward_f16 if sys.argv[1]=="f16" else L.sdfr_mlp_forward
mws = torch.empty(int(L.sdfr_decoder_mask_words(h, inp.shape[0])), dtype=torch.int32, device=dev) if os.environ.get("AB_MASKS") else None
def run(): _lib.check(fn(h,_lib.ptr(inp),inp.shape[0],_lib.ptr(out),_lib.ptr(mws),_lib.stream_ptr()),"fwd")
for _ in range(3): run()
torch.cuda.synchronize(); ts=[]
for r in range(5):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
print("%%.4f %%.4f %%.8f" %% (min(ts), sorted(ts)[2], float(out.double().sum())))
''' % ROOT
prec = sys.argv[1]; B = sys.argv[2]; variants = sys.argv[3:]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, SDFR_MLP_VARIANT=v, SDFR_F16_VARIANT=v)
        out = subprocess.run([sys.executable, "-c", CODE, prec, B], env=env, capture_output=True, text=True)
        print(prec, "B", B, "round", rnd, "variant", v, "min/median ms, checksum:", out.stdout.strip() or out.stderr[-300:], flush=True)
