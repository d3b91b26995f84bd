"""Per-layer cycle trace of one workgroup of the decoder forward kernels (library built with -DSDFR_MLP_TRACE, see tools/ab_variant.sh):
  SDFR_LIB=.../libsdfr_trace.so python tools/cycle_trace.py [f16|f32|split]
Prints, per layer and for waves 0 and 7: product loop, wait at the first barrier, epilogue, wait at the second barrier (shader cycles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, sdflabel_amd
from sdflabel_amd import _lib
from sdflabel_amd.fixtures import ASSET
which = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
h = dec.handle(torch.device(dev, 0)).h
grid = sdflabel_amd.Grid3D(40, dev); lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)
inp = torch.cat([lat.expand(grid.points.size(0), -1), grid.points.detach()], 1).contiguous()
out = torch.empty(inp.shape[0], device=dev)
L = _lib.lib()
mws = torch.empty(int(L.sdfr_decoder_mask_words(h, inp.shape[0])), dtype=torch.int32, device=dev)
fn = {"f16": L.sdfr_mlp_forward_f16, "f32": L.sdfr_mlp_forward, "split": L.sdfr_mlp_forward_split}.get(which)
if which == "t64":
    # the 64-row half forward of the sphere tracer's march (sdfr_mlp_forward_counted, half | 2, on 16 000 rows: one round of 250 tiles)
    inp = inp[:16000].contiguous()
    cnt = torch.tensor([16000], dtype=torch.int32, device=dev)
    fn = lambda h_, i_, n_, o_, m_, s_: L.sdfr_mlp_forward_counted(h_, i_, n_, _lib.ptr(cnt), o_, 3, s_)
trace = torch.zeros(2 * 16 * 5, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.check(fn(h, _lib.ptr(inp), inp.shape[0], _lib.ptr(out), _lib.ptr(mws), _lib.stream_ptr()), "fwd")
L.sdfr_debug_set_trace(_lib.ptr(trace))
_lib.check(fn(h, _lib.ptr(inp), inp.shape[0], _lib.ptr(out), _lib.ptr(mws), _lib.stream_ptr()), "fwd")
torch.cuda.synchronize()
L.sdfr_debug_set_trace(None)
t = trace.cpu().view(2, 16, 5)
print("%s forward, workgroup 0, shader cycles (s_memtime):" % which)
print("layer | wave 0: product  barrier1  epilogue  barrier2 | wave 7: product  barrier1  epilogue  barrier2 | layer total (wave 0)")
tot = 0
for l in range(8):
    row = []
    for w in range(2):
        s = t[w, l]
        row.append((int(s[1] - s[0]), int(s[2] - s[1]), int(s[3] - s[2]), int(s[4] - s[3])))
    lt = int(t[0, l, 4] - t[0, l, 0]); tot += lt
    print("%5d | %15d %9d %9d %9d | %15d %9d %9d %9d | %d" % ((l,) + row[0] + row[1] + (lt,)))
print("sum of the 8 layers (wave 0): %d cycles" % tot)
