"""A/B the single-variant libraries under sdflabel_amd/lib/ab/ (tools/ab_build.sh): decoder forward with mask saving, G = 64000."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os; sys.path.insert(0, %r)
import torch, torch.nn.functional as F, sdflabel_amd
from sdflabel_amd import _lib
from sdflabel_amd.fixtures import ASSET
dev="cuda"; dec,_=sdflabel_amd.setup_dsdf(ASSET+".pt", precision=torch.float32); dec=dec.to(dev)
h = dec.handle(torch.device(dev,0)).h
grid=sdflabel_amd.Grid3D(40,dev); lat=F.normalize(torch.tensor([0.3,-0.5,0.8],device=dev),dim=0)
inp=torch.cat([lat.expand(grid.points.size(0),-1),grid.points],1).contiguous()
out=torch.empty(inp.shape[0],device=dev)
L=_lib.lib()
mws = torch.empty(int(L.sdfr_decoder_mask_words(h, inp.shape[0])), dtype=torch.int32, device=dev)
def run(): _lib.check(L.sdfr_mlp_forward(h,_lib.ptr(inp),inp.shape[0],_lib.ptr(out),_lib.ptr(mws),_lib.stream_ptr()),"fwd")
for _ in range(3): run()
torch.cuda.synchronize(); ts=[]
for r in range(5):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
print("%%.4f %%.4f %%.8f" %% (min(ts), sorted(ts)[2], float(out.double().sum())))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "sdflabel_amd", "lib", "ab", "*.so")))
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in libs:
        env = dict(os.environ, SDFR_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print("round", rnd, os.path.basename(lib), "min/median ms, checksum:", out.stdout.strip() or out.stderr[-300:], flush=True)
