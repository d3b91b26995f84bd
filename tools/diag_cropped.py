"""Diagnostic (development aid): for the G14 cases, the end-to-end gradient deviation of the full +-1-weighted functional against the reference
and the number of pixels whose composite differs from the reference's at several thresholds, overall and among the golden's near-threshold pixels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflabel_amd
from tests._util import ASSET, gold, pattern_weights
from tests.test_gpu_parity import N, T
from tests.test_gpu_configs import SALT
from tests.test_gpu_cropped import Sub
dev = "cuda"
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
dec = dec.to(dev)
for tag in "abc":
    z = Sub(gold("g14_cropped_intrinsics.npz"), tag + "_")
    D, H, W = [int(v) for v in z["cfg"]]
    near = np.unpackbits(z["near_threshold"])[:H * W].astype(bool)
    br = sdflabel_amd.BatchRenderer(dec, D, z["K"], (W, H), 1, device=dev)
    out = br.forward(T(z["yaw"]), T(z["trans"])[None], T(z["latent"])[None])
    nf = z["xyzf"].shape[0]
    gx = torch.zeros(1, br.cap, 3, device=dev); gx[0, :nf] = T(pattern_weights((nf, 3), SALT["xyzf"]))
    w = {k: T(pattern_weights(tuple(out[k].shape[-3:]), SALT[k]))[None] for k in ("color", "mask", "depth", "normals")}
    g = br.backward(g_color=w["color"], g_mask=w["mask"], g_depth=w["depth"], g_normals=w["normals"], g_xyzf=gx)
    dev_rel = [float(np.abs(N(t).reshape(-1) - z["g_" + k].reshape(-1)).max() / max(1.0, np.abs(z["g_" + k]).max())) for t, k in zip(g, ("yaw", "trans", "latent"))]
    d = np.max([np.abs(N(out[k][0]).reshape(-1, H * W) - z["out_" + k].reshape(-1, H * W)).max(0) for k in ("color", "mask", "depth", "normals")], axis=0)
    dn = np.abs(N(br.normals[0, :z["normals"].shape[0]]) - z["normals"]).max(1)
    dp = np.abs(N(br.points[0, :z["pcd"].shape[0]]) - z["pcd"]).max(1)
    print(tag, "N", int(out["n"][0]), "grad dev rel (yaw, trans, latent)", ["%.2e" % v for v in dev_rel], "near px", int(near.sum()), "of", H * W)
    for thr in (1e-4, 1e-5, 1e-6, 3e-7):
        print("   pixels differing >", thr, ":", int((d > thr).sum()), " of them near:", int(((d > thr) & near).sum()))
    print("   surfel normals differing > 1e-4:", int((dn > 1e-4).sum()), "> 1e-6:", int((dn > 1e-6).sum()), " points max", float(dp.max()))
