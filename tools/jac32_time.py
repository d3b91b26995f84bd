"""Time the exact-f32 mask-fed band Jacobian: python tools/jac32_time.py [B ...]   (SDFR_JAC_POOL_CROPS=1000 turns the pool off)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
L = sdflabel_amd._lib.lib(); P = sdflabel_amd._lib.ptr
out = []
for B in [int(a) for a in sys.argv[1:]] or [64]:
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(64, 64), (64, 64), B, device=dev)
    g = torch.Generator().manual_seed(1)
    lat = torch.tensor([[0.3, -0.5, 0.8]]) + 0.2 * (torch.rand(B, 3, generator=g) - 0.5)
    br.set_params(torch.full((B,), 0.7, device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev).expand(B, 3), lat.to(dev))
    br.forward(); torch.cuda.synchronize()
    def jac():
        L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, B, P(br.idx), br.cap, P(br.cnt), P(br.J), P(br.sdf_band), P(br.sdf), P(br.mask_ws), 0,
                            sdflabel_amd._lib.stream_ptr())
    for _ in range(2): jac()
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): jac()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    cs = float(sum(br.J[b, :int(br.cnt[b])].double().sum() for b in range(B)))
    out.append("B=%d: %.1f us (%.1f us/crop, %d rows) checksum %.10g" % (B, min(ts), min(ts) / B, int(br.cnt.sum()), cs))
    del br
print(" | ".join(out))
