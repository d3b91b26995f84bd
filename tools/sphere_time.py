"""Time one sphere-traced render (forward + backward) of the bench crop: python tools/sphere_time.py [f16|f32] [steps] (development aid;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sdflabel_amd
from sdflabel_amd.fixtures import ASSET
prec = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.float32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
d3, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
tr = sdflabel_amd.SphereTracer(d3.to(dev), bench.K_for(bench.H, bench.W), (bench.W, bench.H), 1, steps=steps, device=dev)
crop = bench.Crop(0, dev)
prm = [crop.yaw.detach().clone().requires_grad_(True), crop.trans.detach().clone().view(1, 3).requires_grad_(True),
       crop.latent.detach().clone().view(1, -1).requires_grad_(True)]
def tstep(bwd=True):
    for p_ in prm:
        p_.grad = None
    o_ = tr(*prm)
    if bwd:
        (o_["depth"].sum() + o_["color"].sum() + o_["normals"].sum()).backward()
for _ in range(2):
    tstep()
torch.cuda.synchronize()
for bwd in (True, False):
    t = time.perf_counter()
    for _ in range(5):
        tstep(bwd)
    torch.cuda.synchronize()
    print("%s %d steps (%d run), backward %s: %.2f ms per render" % (sys.argv[1] if len(sys.argv) > 1 else "f32", steps, tr.steps_run, bwd, (time.perf_counter() - t) / 5 * 1e3))
# the march alone, and the active-ray count after every step (one synchronisation per step: timing not representative)
R = sdflabel_amd.renderer.sphere_tracer
with torch.no_grad():
    yaw, trans, lat = prm
    Rm = R._rot_from_yaw(yaw.reshape(1)); latn = torch.nn.functional.normalize(lat, p=2, dim=1)
    pose = torch.zeros(1, 4, 4, device=dev); pose[:, :3, :3] = Rm; pose[:, :3, 3] = trans; pose[:, 3, 3] = 1.0
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        tr.march(pose.view(1, 16), latn)
    torch.cuda.synchronize()
    print("march alone: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
    ce, tr.check_every = tr.check_every, 1
    tr.stop_fraction_saved, tr.stop_fraction = tr.stop_fraction, -1.0
    counts = []
    import types
    # active counts: run with increasing step limits
    for s in (1, 2, 4, 8, 12, 16, 24, 32, 40, 48, 56, 64):
        if s > steps: break
        tr.steps = s
        tr.march(pose.view(1, 16), latn)
        counts.append((s, int(tr.n_unresolved)))
    print("active rays after k steps:", counts)
