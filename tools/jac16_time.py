"""Time the half-operand mask-fed band Jacobian of the float16 decoder: python tools/jac16_time.py [B ...]  (SDFR_LIB selects a variant)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); dec = dec.to(dev)
L = sdflabel_amd._lib.lib(); P = sdflabel_amd._lib.ptr
out = []
FLAG = int(os.environ.get("SDFR_JAC_FLAG", "2"))     # 2: half backward on 16-row tiles; 18 (= 2 | SDFR_JAC_MANY_ROWS): 32x32 tiles
for B in [int(a) for a in sys.argv[1:]] or [int(os.environ.get("SDFR_JAC_B", "64"))]:
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(64, 64), (64, 64), B, device=dev)
    g = torch.Generator().manual_seed(1)
    lat = torch.tensor([[0.3, -0.5, 0.8]]) + 0.2 * (torch.rand(B, 3, generator=g) - 0.5)
    br.set_params(torch.full((B,), 0.7, device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev).expand(B, 3), lat.to(dev))
    br.forward(); torch.cuda.synchronize()
    def jac():
        L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, B, P(br.idx), br.cap, P(br.cnt), P(br.J), P(br.sdf_band), P(br.sdf), P(br.mask_ws), FLAG,
                            sdflabel_amd._lib.stream_ptr())
    for _ in range(3): jac()
    n = 50 if B < 16 else 10
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): jac()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    cs = float(sum(br.J[b, :int(br.cnt[b])].double().sum() for b in range(B)))
    out.append("B=%d: %.1f us (%.1f us/crop) checksum %.8g" % (B, min(ts), min(ts) / B, cs))
    del br
print(os.path.basename(os.environ.get("SDFR_LIB", "default")), "flag", FLAG, " | ".join(out))
if os.environ.get("SDFR_JAC_TRACE"):
    # per-layer cycle stamps of workgroup (0, 0) (library built with SDFR_J16_DEFS=-DSDFR_MLP_TRACE): product loop incl. mask fetch, wait at the
    # first barrier, mask application + operand store, wait at the second barrier; waves 0 and 7
    B = int(os.environ.get("SDFR_JAC_B", "64"))
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(64, 64), (64, 64), B, device=dev)
    br.set_params(torch.full((B,), 0.7, device=dev), torch.tensor([[0.05, 0.02, 3.3]], device=dev).expand(B, 3), torch.tensor([[0.3, -0.5, 0.8]], device=dev).expand(B, 3))
    br.forward(); torch.cuda.synchronize()
    trace = torch.zeros(2 * 16 * 5, dtype=torch.int64, device=dev)
    L.sdfr_debug_set_trace(P(trace))
    L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, B, P(br.idx), br.cap, P(br.cnt), P(br.J), P(br.sdf_band), P(br.sdf), P(br.mask_ws), FLAG, sdflabel_amd._lib.stream_ptr())
    torch.cuda.synchronize()
    L.sdfr_debug_set_trace(None)
    t = trace.cpu().view(2, 16, 5)
    print("layer | wave 0: product  barrier1  epilogue  barrier2 | wave 7: product  barrier1  epilogue  barrier2 | layer total (wave 0)")
    for l in range(7, 0, -1):
        row = []
        for w in range(2):
            s = t[w, l]
            row.append((int(s[1] - s[0]), int(s[2] - s[1]), int(s[3] - s[2]), int(s[4] - s[3])))
        print("%5d | %15d %9d %9d %9d | %15d %9d %9d %9d | %d" % ((l,) + row[0] + row[1] + (int(t[0, l, 4] - t[0, l, 0]),)))
    print("layers 7..1 (wave 0): %d cycles" % int(t[0, 1, 4] - t[0, 7, 0]))
    for w in range(2):
        h, z = t[w, 8], t[w, 9]
        print("wave %d: rows/slots %d | J zero + operand %d | output gradient %d | top in-gradient (w_last x gy, masks of the last layer) %d | layers 7..1 %d | layer 0: VALU product %d, reduction + atomics %d | whole tile %d"
              % (7 * w, int(h[1] - h[0]), int(h[2] - h[1]), int(h[3] - h[2]), int(h[4] - h[3]), int(z[0] - h[4]), int(z[1] - z[0]), int(z[2] - z[1]), int(z[2] - h[0])))
