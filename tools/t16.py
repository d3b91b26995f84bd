import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev="cuda"
dec,_=sdflabel_amd.setup_dsdf(ASSET+".pt", precision=torch.float16); dec=dec.to(dev)
grid=sdflabel_amd.Grid3D(40,dev); lat=F.normalize(torch.tensor([0.3,-0.5,0.8],device=dev),dim=0)
inp=torch.cat([lat.expand(grid.points.size(0),-1),grid.points.detach()],1).contiguous()
with torch.no_grad():
    for _ in range(3): s,_=dec(inp)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): s,_=dec(inp)
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("SDFR_LIB","default"), "%.4f ms" % (e0.elapsed_time(e1)/20), float(s.double().sum()))
