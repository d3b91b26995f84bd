"""The reference's SHIPPED operating point (configs/config_refine.ini:11-19: grid_density 40, rendering_area 32, iters 60, precision float16; one
Optimizer per annotation, pipelines/refine_css.py:94,203-223) timed three ways on KITTI-like crops (sdflabel_amd.fixtures.kitti_like_problems):

  per_annotation   Optimizer(params, device, weights).optimize(60, ...) per crop, as the pipeline issues them
  phases           the same call taken apart on ONE crop: set_crops / enqueue of 60 graph replays / GPU time of the replays / check_overflow
  optimize_many    Optimizer.optimize_many(annotations of a frame, ...) for frames of 4 / 8 / 16 / 32 annotations
  identical        optimize_many results == per-annotation results, bit for bit

    python tools/area32_time.py [--area 32] [--n 32] [--precision float16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, kitti_like_problems
from sdflabel_amd.pipelines import optimizer as OP


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--area", type=int, default=32)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--precision", default="float16", choices=["float16", "float32"])
    ap.add_argument("--only", default="", help="comma list of sections (per_annotation,phases,many); default all")
    ap.add_argument("--frames", default="4,8,16,32")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--set", default="", help="comma list attr=0/1 set on the decoder (e.g. candidate_quarter_tiles=1,fused_launches=0)")
    ap.add_argument("--hostprof", action="store_true", help="cProfile of the per-annotation loop (host side)")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
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
    dec.latent_lipschitz_bound()
    shapes, Ks, targets, lidars, starts = kitti_like_problems(dec32, D, args.area, args.n, dev)
    grid = sdflabel_amd.Grid3D(D, dev)
    W8 = {"2d": 0.3, "3d": 0.5}
    out = {"area": args.area, "n": args.n, "precision": args.precision, "crop_sizes_min_max": [list(min(shapes)), list(max(shapes))]}

    def fresh(b):
        return {k: v.copy() for k, v in starts[b].items()}

    def per_annotation(n):
        res = []
        for b in range(n):
            opt = OP.Optimizer(fresh(b), dev, W8)
            res.append(opt.optimize(args.iters, targets[b], lidars[b], dec, grid, torch.from_numpy(Ks[b]), list(shapes[b])))
        return res

    single = None
    if not only or "per_annotation" in only:
        per_annotation(2)                                   # builds + captures the refiner
        best = None
        for _ in range(args.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            single = per_annotation(args.n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["per_annotation"] = {"crops_per_s": args.n / best, "ms_per_crop": best / args.n * 1e3}
        print("per_annotation", json.dumps(out["per_annotation"]), flush=True)

    if args.hostprof:
        import cProfile, pstats, io
        per_annotation(2)
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        per_annotation(args.n)
        torch.cuda.synchronize()
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
        print(st.getvalue())

    if not only or "phases" in only:
        opt = OP.Optimizer(fresh(0), dev, W8)
        opt.optimize(args.iters, targets[0], lidars[0], dec, grid, torch.from_numpy(Ks[0]), list(shapes[0]))
        rf = opt._refiner
        ph = {}
        P1 = {k: torch.as_tensor(v).reshape(1, -1) for k, v in starts[0].items()}

        def sc():
            rf.set_crops(P1, [targets[0]], [lidars[0]], K=Ks[0], crop_sizes=[shapes[0]])
        for name, fn in (("set_crops", sc), ("enqueue_60_replays", lambda: rf.optimize(args.iters)), ("check_overflow", rf.check_overflow)):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
            ph[name] = {"host_ms": float(np.median([a for a, _ in ts])), "host_plus_gpu_ms": float(np.median([b for _, b in ts]))}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sc()
        torch.cuda.synchronize()
        e0.record()
        rf.optimize(args.iters)
        e1.record()
        torch.cuda.synchronize()
        ph["gpu_ms_per_iteration"] = e0.elapsed_time(e1) / args.iters
        ph["surfels"] = int(rf.br.cnt[0]); ph["front_facing"] = int(rf.br.fcnt[0]); ph["candidates"] = int(rf.br.ccnt[0]) if rf.br.creuse else None
        out["phases"] = ph
        print("phases", json.dumps(ph), flush=True)

    if not only or "many" in only:
        many = {}
        for fsz in [int(x) for x in args.frames.split(",")]:
            if fsz > args.n:
                continue

            def frame(c0):
                return [(fresh(b), targets[b], lidars[b], Ks[b], shapes[b]) for b in range(c0, min(args.n, c0 + fsz))]
            OP.optimize_many(frame(0), args.iters, dec, grid, dev, W8)            # build + capture
            best, res = None, None
            for _ in range(args.repeat):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = []
                for c0 in range(0, args.n, fsz):
                    res += OP.optimize_many(frame(c0), args.iters, dec, grid, dev, W8)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            many["frame_of_%d" % fsz] = {"crops_per_s": args.n / best, "ms_per_frame": best / ((args.n + fsz - 1) // fsz) * 1e3}
            if single is not None:
                same = all(torch.equal(res[b][k], single[b][k]) for b in range(args.n) for k in ("yaw", "trans", "scale", "latent"))
                many["frame_of_%d" % fsz]["identical_to_per_annotation"] = bool(same)
            print("many", fsz, json.dumps(many["frame_of_%d" % fsz]), flush=True)
        out["optimize_many"] = many
    print(json.dumps(out))


if __name__ == "__main__":
    main()
