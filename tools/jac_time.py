"""Time the mask-fed band Jacobian alone (after one decoder forward): python tools/jac_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflabel_amd
from tests._util import ASSET, K_for
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
br = sdflabel_amd.BatchRenderer(dec, 40, K_for(256, 256), (256, 256), 1, device=dev)
br.set_params(torch.tensor([0.7], device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
br.forward(); torch.cuda.synchronize()
L = sdflabel_amd._lib.lib(); P = sdflabel_amd._lib.ptr
def jac():
    L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, 1, P(br.idx), br.cap, P(br.cnt), P(br.J), P(br.sdf_band), P(br.sdf), P(br.mask_ws), 0,
                        sdflabel_amd._lib.stream_ptr())
for _ in range(5): jac()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): jac()
e1.record(); torch.cuda.synchronize()
print("jacobian %.1f us  (N = %d rows)" % (e0.elapsed_time(e1) / 50 * 1e3, int(br.cnt[0])))
