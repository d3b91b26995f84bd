"""Time the splat forward / backward (and the fused surfel-forward kernel that bins) of the bench workload with events on the launch stream:
python tools/splat_time.py [B] [crop]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflabel_amd
from sdflabel_amd import _lib
from sdflabel_amd.fixtures import ASSET, K_for
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
br = sdflabel_amd.BatchRenderer(dec, 40, K_for(HW, HW), (HW, HW), B, device=dev)
g = torch.Generator().manual_seed(1)
yaw = 0.7 + 0.1 * torch.rand(B, generator=g); trans = torch.tensor([[0.0, 0.0, 3.5]]) + torch.rand(B, 3, generator=g) * torch.tensor([[0.1, 0.05, -0.3]])
lat = torch.tensor([[0.3, -0.5, 0.8]]) + 0.2 * (torch.rand(B, 3, generator=g) - 0.5)
br.set_params(yaw.to(dev), trans.to(dev), lat.to(dev))
ones3 = torch.ones(B, 3, HW, HW, device=dev); ones1 = torch.ones(B, 1, HW, HW, device=dev); onesx = torch.ones(B, br.cap, 3, device=dev)
for _ in range(3):
    br.forward(); br.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)
torch.cuda.synchronize()
L = _lib.lib(); P = _lib.ptr
def timeit(fn, n=50):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]
W = H = HW
xyz = br.inputs[:, br.NI - 3:]
def surf():
    _lib.check(L.sdfr_surfels_forward(P(xyz), br.NI, P(br.sdf), br.G, P(br.idx), P(br.J), br.NI, br.NI - 3, P(br.pose), P(br.K), B, br.cap, P(br.cnt), br.nocs_mode | 4 | 8, W, H, 0.04,
               P(br.points), P(br.normals), P(br.p_cam), P(br.n_cam), P(br.attr), P(br.fidx), P(br.fcnt), P(br.xyzf), P(br.fslot), P(br.bbox), _lib.stream_ptr()), "surf")
def splat(flags):
    def f():
        _lib.check(L.sdfr_splat_forward(flags, P(br.K), P(br.Kinv), P(br.p_cam), P(br.n_cam), P(br.attr), None, None, None, None, B, br.cap, P(br.cnt), W, H, 0.04, 150.0,
                   P(br.bbox), P(br.color), P(br.mask), P(br.depth), P(br.nimg), P(br.aux), _lib.stream_ptr()), "splat")
    return f
def bwd():
    _lib.check(L.sdfr_splat_backward(0, P(br.K), P(br.Kinv), P(br.p_cam), P(br.n_cam), P(br.attr), None, None, None, None, B, br.cap, P(br.cnt), W, H, 0.04, 150.0, P(br.aux),
               P(br.color), P(br.mask), P(br.depth), P(br.nimg), P(ones3), P(ones1), None, P(ones3), P(br.g_p), P(br.g_n), P(br.g_a), _lib.stream_ptr()), "bwd")
n, nf = int(br.cnt[0]), int(br.fcnt[0])
print("B=%d %dx%d N=%d Nf=%d" % (B, HW, HW, n, nf))
print("surfels_forward (projection + boxes + bins)  median %.1f us  min %.1f" % timeit(surf))
print("splat fwd, lists ready (256|512)              median %.1f us  min %.1f" % timeit(splat(256 | 512)))
print("splat fwd, boxes+bins rebuilt (512)           median %.1f us  min %.1f" % timeit(splat(512)))
print("splat fwd, unbinned scan (256)                median %.1f us  min %.1f" % timeit(splat(256)))
print("splat bwd                                     median %.1f us  min %.1f" % timeit(bwd))
surf()
