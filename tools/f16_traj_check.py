"""Which part of the float16 set-up moves the refinement trajectory away from the reference's (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflabel_amd
from sdflabel_amd.pipelines.optimizer import Optimizer
from sdflabel_amd.fixtures import ASSET
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z, zh = np.load(os.path.join(G, "g8_optimizer.npz")), np.load(os.path.join(G, "g8h_optimizer_fp16.npz"))
DEV = "cuda"; D, H, W = int(z["D"]), int(z["H"]), int(z["W"]); init = z["init"]
T = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
def run(dec_prec, io_half, half_jac=None):
    dsdf, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=dec_prec); dsdf = dsdf.to(DEV)
    if half_jac is not None:
        dsdf.half_jacobian = half_jac
    grid = sdflabel_amd.Grid3D(D, DEV, torch.float16 if io_half else torch.float32)
    K = T(z["K"]); nocs = T(z["nocs_target"])
    if io_half: K, nocs = K.half(), nocs.half()
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    traj = []
    for _ in range(10):
        opt.optimize(1, nocs, z["lidar"], dsdf, grid, K, (H, W))
        traj.append(np.concatenate([params[k].detach().cpu().numpy().reshape(-1) for k in ("yaw", "trans", "scale", "latent")]))
    traj = np.asarray(traj)
    return np.abs(traj - z["traj"]).max(0)[:5], np.abs(traj - zh["traj"]).max(0)[:5]
np.set_printoptions(precision=2, suppress=False)
for name, a in (("f32 decoder, f32 io", (torch.float32, False)), ("f32 decoder, half io", (torch.float32, True)),
                ("f16 decoder, f32 io", (torch.float16, False)), ("f16 decoder, half io", (torch.float16, True)),
                ("split decoder, f32 io", ("float32_split", False))):
    e32, e16 = run(*a)
    print("%-24s vs ref f32 %s   vs ref f16 %s" % (name, e32, e16))
