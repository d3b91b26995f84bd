import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_params, synthetic_targets, crop_start
from sdflabel_amd.pipelines import optimizer as OP
dev = "cuda"; D = 40; iters = int(sys.argv[2]); area = 256; render = sys.argv[1]
dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32); dec = dec.to(dev)
d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16); d16 = d16.to(dev)
rng = np.random.default_rng(11)
n = 32
boxes_w = rng.uniform(60, 420, n)
boxes_h = boxes_w / rng.uniform(1.2, 3.2, n)
shapes, Ks, gts = [], [], []
for bw, bh in zip(boxes_w, boxes_h):
    r = np.sqrt(area * area / (bh * bw))
    Hc, Wc = int(bh * r), int(bw * r)
    f = 1.15 * Hc * 3.5 / 2.0
    cx, cy = rng.uniform(-1.0 * Wc, 2.0 * Wc), rng.uniform(0.2 * Hc, 0.8 * Hc)
    shapes.append((Hc, Wc))
    Ks.append(np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], np.float32))
    gts.append(np.array([3.5 * (Wc / 2.0 - cx) / f, 3.5 * (Hc / 2.0 - cy) / f, 3.5], np.float32))
pmax = 1 << (max(h * w for h, w in shapes) - 1).bit_length()
gtr = sdflabel_amd.BatchRenderer(dec, D, np.stack(Ks), (shapes[0][1], shapes[0][0]), n, device=dev, max_pixels=pmax)
gtr.set_extents([(w, h) for h, w in shapes], np.stack(Ks))
o = gtr.forward(torch.full((n,), 0.6, device=dev), torch.from_numpy(np.stack(gts)).to(dev), torch.tensor([[0.3, -0.5, 0.8]] * n, device=dev))
nfs = o["nf"].tolist()
targets = [gtr.image(b, "color").clone().cpu() for b in range(n)]
lidars = [(o["xyzf"][b, :nfs[b]] * 2.0)[::2].cpu().numpy() for b in range(n)]
del gtr
grid = sdflabel_amd.Grid3D(D, dev)
starts = [crop_start(i) for i in range(n)]
torch.cuda.synchronize()
for b in range(n):
    y0, t0_, l0 = starts[b]
    p = {"yaw": y0.copy(), "trans": (gts[b] + (t0_ - np.asarray([0.0, 0.0, 3.5], np.float32))).astype(np.float32), "scale": np.array([2.0], np.float32), "latent": l0.copy()}
    opt = OP.Optimizer(p, dev, {"2d": 0.3, "3d": 0.5}, render=render)
    out = opt.optimize(iters, targets[b], lidars[b], d16, grid, torch.from_numpy(Ks[b]), list(shapes[b]))
    torch.cuda.synchronize()
    print(b, shapes[b], len(lidars[b]), float(out["yaw"][0]), flush=True)
print("done")
