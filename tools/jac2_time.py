"""Time the recomputing Jacobian (MODE 2: exact-f32 forward + backward of selected rows, no saved masks): python tools/jac2_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
br = sdflabel_amd.BatchRenderer(dec, 40, K_for(256, 256), (256, 256), 1, device=dev)
br.set_params(torch.tensor([0.7], device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
br.forward(); torch.cuda.synchronize()
L = sdflabel_amd._lib.lib(); P = sdflabel_amd._lib.ptr
# candidates: |sdf| < 0.04
idx = torch.nonzero(br.sdf.abs() < 0.04).view(-1).to(torch.int32)
n = int(idx.numel()); cap = br.cap
idxb = torch.zeros(cap, dtype=torch.int32, device=dev); idxb[:n] = idx
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
J = torch.empty(cap, br.NI, device=dev); sel = torch.empty(cap, device=dev)
def jac():
    L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, 1, P(idxb), cap, P(cnt), P(J), P(sel), None, None, 0, sdflabel_amd._lib.stream_ptr())
for _ in range(5): jac()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): jac()
e1.record(); torch.cuda.synchronize()
print("recomputing jacobian %.1f us  (%d candidate rows; band %d)  max|sdf_sel - sdf| %.2e" %
      (e0.elapsed_time(e1) / 50 * 1e3, n, int(br.cnt[0]), float((sel[:n] - br.sdf[idx.long()]).abs().max())))
