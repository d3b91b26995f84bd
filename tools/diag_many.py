"""Where does a crop refined in a batch of B start to differ from the same crop refined alone?  (r06 diagnostic: optimize_many vs Optimizer)

    python tools/diag_many.py [--area 32] [--B 4] [--precision float16] [--no-reuse] [--iters 60]

Two ragged BatchRefiners (batch 1 and batch B) on the same KITTI-like crops; after every iteration the arrays of crop 0 are compared and the
first differing one is printed with its iteration.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, kitti_like_problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--area", type=int, default=32)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--precision", default="float16")
    ap.add_argument("--no-reuse", action="store_true")
    ap.add_argument("--crop", type=int, default=0)
    ap.add_argument("--binned", default="", help="two characters 0/1: force BatchRenderer.binned of the batch-1 and the batch-B refiner")
    ap.add_argument("--set", default="", help="comma list attr=value applied to BOTH decoders (e.g. candidate_half_tiles=0)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    D = 40
    dec32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    dec32 = dec32.to(dev)
    prec = torch.float16 if args.precision == "float16" else torch.float32
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
    dec = dec.to(dev)
    for kv in [x for x in args.set.split(",") if x]:
        k, v = kv.split("=")
        setattr(dec, k, bool(int(v)))
    n = max(args.B, args.crop + 1)
    shapes, Ks, targets, lidars, starts = kitti_like_problems(dec32, D, args.area, n, dev)
    pmax = max(1024, 1 << (max(h * w for h, w in shapes) - 1).bit_length())
    side = 4 * int(np.ceil(np.sqrt(pmax)))
    c = args.crop
    order = [c] + [i for i in range(args.B) if i != c][:args.B - 1]

    def make(B, ids):
        rf = sdflabel_amd.BatchRefiner(dec, D, Ks[ids[0]], shapes[ids[0]], B, lidar_cap=4096, device=dev, max_pixels=pmax, max_side=side,
                                       candidate_reuse=not args.no_reuse)
        P = {k: np.stack([starts[i][k].reshape(-1) for i in ids]) for k in ("yaw", "trans", "scale", "latent")}
        rf.set_crops(P, [targets[i] for i in ids], [lidars[i] for i in ids], K=np.stack([Ks[i] for i in ids]), crop_sizes=[shapes[i] for i in ids])
        return rf
    r1, rB = make(1, [c]), make(args.B, order)
    if args.binned:
        r1.br.binned, rB.br.binned = args.binned[0] == "1", args.binned[1] == "1"
    print("crop %d: H x W = %s" % (c, shapes[c]))
    names = ["K", "Kinv", "cnt", "ccnt", "idx", "sdf_band", "J", "points", "normals", "p_cam", "n_cam", "attr", "boxes", "fcnt", "xyzf", "aux", "mask", "depth", "nimg", "color", "loss2d", "loss3d", "g_color", "g_xyzf", "grads", "params"]

    def arrays(rf):
        br = rf.br
        nb = int(br.cnt[0])
        nf = int(br.fcnt[0])
        w, h = br.sizes[0]
        B = rf.B
        g = rf.grads
        p = rf.params
        sec = lambda buf: torch.cat([buf[0:1], buf[B:B + 3], buf[4 * B:4 * B + 1], buf[5 * B:5 * B + rf.L]])
        return {"K": br.K[0], "Kinv": br.Kinv[0], "p_cam": br.p_cam[0, :nb], "n_cam": br.n_cam[0, :nb], "attr": br.attr[0, :nb], "boxes": br.boxes[0, :nb],
                "aux": br.aux[0, :w * h], "mask": br.mask[0, :, :w * h], "depth": br.depth[0, :, :w * h], "nimg": br.nimg[0, :, :w * h],
                "cnt": br.cnt[0:1], "ccnt": br.ccnt[0:1] if br.creuse else br.cnt[0:1], "idx": br.idx[0, :nb], "sdf_band": br.sdf_band[0, :nb], "J": br.J[0, :nb],
                "points": br.points[0, :nb], "normals": br.normals[0, :nb], "fcnt": br.fcnt[0:1], "xyzf": br.xyzf[0, :nf], "color": br.color[0, :, :w * h],
                "loss2d": rf.loss2d[0:1], "loss3d": rf.loss3d[0:1], "g_color": rf.g_color[0, :, :w * h], "g_xyzf": rf.g_xyzf[0, :nf], "grads": sec(g),
                "params": sec(p)}
    for it in range(args.iters):
        r1.iteration(); rB.iteration()
        a, b = arrays(r1), arrays(rB)
        w_, h_ = r1.br.sizes[0]
        bad = [k for k in names if a[k].shape != b[k].shape or not torch.equal(a[k], b[k])]
        if bad:
            k = bad[0]
            d = (a[k].float() - b[k].float()).abs().max().item() if a[k].shape == b[k].shape else float("nan")
            print("iteration %d: first differing array %r (max abs diff %g); all differing: %s" % (it, k, d, bad))
            if k in ("color", "aux", "mask", "depth", "nimg"):
                dd = (a[k].float() - b[k].float()).abs().reshape(-1, w_ * h_ if k != "aux" else 4)
                if k != "aux":
                    px = torch.nonzero(dd.sum(0) > 0).reshape(-1).tolist()
                    print("  differing pixels (x, y):", [(q % w_, q // w_) for q in px][:20], "of", w_, "x", h_)
            print("  Kinv B=1:", a["Kinv"].reshape(-1).tolist())
            print("  Kinv B=n:", b["Kinv"].reshape(-1).tolist())
            print("  B=1 flags: binned %s half_tiles %s audit_side %s | B=%d flags: binned %s half_tiles %s audit_side %s" % (
                r1.br.binned, getattr(r1.br, "half_tiles", None), getattr(r1.br, "audit_side", None), args.B, rB.br.binned,
                getattr(rB.br, "half_tiles", None), getattr(rB.br, "audit_side", None)))
            return 1
    print("identical over %d iterations (crop %d alone vs first of %d)" % (args.iters, c, args.B))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
