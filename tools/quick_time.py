"""Ad-hoc kernel timing on the GPU box (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import sdflabel_amd
from tests._util import ASSET, K_for

dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt"); dec = dec.to(dev)
D = 40
grid = sdflabel_amd.Grid3D(D, dev)
lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)
inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1).contiguous()

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

with torch.no_grad():
    t = timeit(lambda: dec(inputs))
G = inputs.shape[0]
macs = dec.handle(torch.device(dev, 0)).macs
print("mlp_forward G=%d: %.3f ms  -> %.1f TFLOP/s (f32 MFMA peak 157.3)" % (G, t, 2 * macs * G / t / 1e9))
for B in (4, 16):
    big = inputs.repeat(B, 1).contiguous()
    with torch.no_grad():
        t = timeit(lambda: dec(big), n=5)
    print("mlp_forward G=%d: %.3f ms  -> %.1f TFLOP/s" % (big.shape[0], t, 2 * macs * big.shape[0] / t / 1e9))

H = W = 256
r = sdflabel_amd.Rasterer(torch.from_numpy(K_for(H, W)), (W, H)).to(dev)
def step():
    l = torch.tensor([0.3, -0.5, 0.8], device=dev, requires_grad=True)
    yaw = torch.tensor([0.6], device=dev, requires_grad=True)
    trans = torch.tensor([0.0, 0.0, 3.5], device=dev, requires_grad=True)
    l_ = F.normalize(l, p=2, dim=0)
    inp = torch.cat([l_.expand(G, -1), grid.points], 1)
    sdf, _ = dec(inp)
    pcd, _, nrm = grid.get_surface_points(sdf)
    c, s = torch.cos(yaw), torch.sin(yaw); z, o = yaw.new_zeros(1), yaw.new_ones(1)
    pose = torch.eye(4, device=dev); pose[:3, :3] = torch.stack((c, z, s, z, o, z, -s, z, c)).view(3, 3); pose[1] *= -1; pose[:3, 3] = trans
    rend, pts = r(pcd, nrm, nrm, pose, rot="dcm", output_mask=True, output_normals=True, output_nocs=True)
    (rend["color"].sum() + rend["normals"].sum() + pts["xyzf"].sum()).backward()
    return pcd.shape[0]
n = step()
t = timeit(step, n=10)
print("drop-in crop-iteration 256x256 D=40 N=%d: %.3f ms -> %.2f Mrays/s" % (n, t, H * W / t / 1e3))
