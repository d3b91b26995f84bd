"""Accuracy and speed of the three decoder forwards (exact f32 MFMA, error-compensated f16 pairs, plain f16) against a float64
evaluation of the same network on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import sdflabel_amd
from sdflabel_amd.fixtures import ASSET
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
grid = sdflabel_amd.Grid3D(40, dev)
lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)
inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points.detach()], 1).contiguous()

def f64_forward(x):
    layers = dec.effective_layers()
    inj = dec._inject_table()
    x0 = x.astype(np.float64); h = x0
    for l, (W, b) in enumerate(layers):
        if inj[l][0]:
            h = np.concatenate([h, x0[:, inj[l][1]:inj[l][1] + inj[l][0]]], 1)
        h = h @ W.astype(np.float64).T + b.astype(np.float64)
        if l < len(layers) - 1:
            h = np.maximum(h, 0)
    return np.tanh(h[:, 0])

sel = torch.arange(0, inputs.shape[0], 7, device=dev)
ref = f64_forward(inputs[sel].cpu().numpy())
out = {}
for name, prec in (("f32", torch.float32), ("split", "float32_split"), ("f16", torch.float16)):
    dec.mlp_precision = prec
    with torch.no_grad():
        sdf, _ = dec(inputs)
        st = sdf._sdfr_state
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L = sdflabel_amd._lib.lib()
        fwd = {"f32": L.sdfr_mlp_forward, "split": L.sdfr_mlp_forward_split, "f16": L.sdfr_mlp_forward_f16}[name]
        P = sdflabel_amd._lib.ptr
        o = torch.empty_like(st.sdf)
        for _ in range(3):
            fwd(st.handle.h, P(st.inputs), st.G, P(o), P(st.mask_ws), sdflabel_amd._lib.stream_ptr())
        e0.record()
        for _ in range(20):
            fwd(st.handle.h, P(st.inputs), st.G, P(o), P(st.mask_ws), sdflabel_amd._lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
    s = sdf.view(-1)
    out[name] = (s.clone(), st.mask_ws.clone())
    err = np.abs(s[sel].cpu().numpy().astype(np.float64) - ref)
    band = int((s.abs() < 0.03).sum())
    print("%-5s  %.3f ms  %.0f TFLOP/s(alg)  max|err vs f64| %.3e  mean %.3e  band %d" %
          (name, ms, 2.0 * st.handle.macs * st.G / ms / 1e9, err.max(), err.mean(), band))
for name in ("split", "f16"):
    d = (out[name][0] - out["f32"][0]).abs()
    n = min(out[name][1].numel(), out["f32"][1].numel())
    flips = "n/a (different layout)" if name == "f16" else str(int(torch.count_nonzero(out[name][1][:n] ^ out["f32"][1][:n])))
    bx = ((out[name][0].abs() < 0.03) != (out["f32"][0].abs() < 0.03)).sum().item()
    print("%-5s vs f32: max|dsdf| %.3e  band membership differs at %d rows  mask words differing %s" % (name, d.max().item(), bx, flips))
