"""A/B of the 64-row half kernels (tools/ab_t64.sh variants under sdflabel_amd/lib/ab/): latency of one decoder pass over n rows through
sdfr_mlp_forward_counted(half | 2) -- the tracer's thin passes -- and a default sphere-tracer render, per library (SDFR_LIB)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os; sys.path.insert(0, %r)
import numpy as np, torch, sdflabel_amd
from sdflabel_amd import _lib
from sdflabel_amd.fixtures import ASSET, K_for, crop_start
dev = "cuda"
d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); d = d.to(dev)
h = d.handle(torch.device(dev, 0)).h
L = _lib.lib()
out = []
for n in (1024, 4096, 16383, 32768):
    inp = torch.randn(n, 6, device=dev) * 0.5
    sdf = torch.empty(n, device=dev); cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    run = lambda: _lib.check(L.sdfr_mlp_forward_counted(h, _lib.ptr(inp), n, _lib.ptr(cnt), _lib.ptr(sdf), 3, _lib.stream_ptr()), "f")
    for _ in range(5): run()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%%d rows %%.1f us" %% (n, min(ts)))
H = W = 256
st = crop_start(0)
prm = [torch.tensor(st[0], device=dev), torch.tensor(st[1][None], device=dev), torch.tensor(st[2][None], device=dev)]
tr = sdflabel_amd.SphereTracer(d, K_for(H, W), (W, H), 1, steps=64, device=dev)
o3, o1 = torch.ones(1, 3, H, W, device=dev), torch.ones(1, 1, H, W, device=dev)
def step():
    tr.render(*prm); tr.backward(g_color=o3, g_depth=o1, g_normals=o3)
for _ in range(5): step()
torch.cuda.synchronize(); import time; ts = []
for r in range(5):
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
print(" | ".join(out), "| render fwd+bwd %%.3f ms (hits %%d, checksum %%.6f)" %% (min(ts), tr.n_hit, float(tr.depth.double().sum())))
''' % ROOT
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "sdflabel_amd", "lib", "ab", "libsdfr_t64_*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["SDFR_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) if lib else "default (PF 2, PFB 2)", ":", out.stdout.strip() or out.stderr[-400:], flush=True)
