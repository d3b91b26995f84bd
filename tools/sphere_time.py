"""Timing of the sphere-tracing render mode on the GPU box: one 256x256 (or --size) crop, forward + backward, per decoder precision and march
schedule: python tools/sphere_time.py [--size 256] [--steps 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sdflabel_amd
from sdflabel_amd.fixtures import ASSET, K_for, crop_start

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--spec", type=int, nargs="*", default=[1, 4], help="samples per ray and pass in the looping kernel")
ap.add_argument("--scan", default="", help="'from:from2,from:from2,...' -- schedules to time instead of the built-in list")
ap.add_argument("--cone", type=int, nargs="*", default=[-1], help="cone_block values to time (0: no cone phase; -1: the tracer's default, 4)")
ap.add_argument("--cone-steps", type=int, default=None, help="cone passes (default: the tracer's own, 4 with 4 samples per pass)")
ap.add_argument("--only", default="", help="f32 | f16: one precision, default schedule only (profiling runs)")
ap.add_argument("--kw", default="", help="JSON list of SphereTracer keyword dicts to time instead of the built-in schedules, e.g. '[{\"tail_rows\": 0}, {\"uniform_tiles\": false}]'")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = "cuda"
H = W = args.size
B = args.batch
st = [crop_start(i) for i in range(B)]
prm = [torch.tensor(np.concatenate([s[0] for s in st]), device=dev), torch.tensor(np.stack([s[1] for s in st]), device=dev),
       torch.tensor(np.stack([s[2] for s in st]), device=dev)]
o3, o1 = torch.ones(B, 3, H, W, device=dev), torch.ones(B, 1, H, W, device=dev)
for prec in ((torch.float32, torch.float16) if not args.only else ((torch.float16,) if args.only == "f16" else (torch.float32,))):
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
    d = d.to(dev)
    macs = d.handle(torch.device(dev)).macs
    scheds = [dict(spec_k=k) for k in args.spec]                              # (spec_k given: the r03 two-level schedule, q_max 1)
    if args.only:
        scheds = [dict()]                                                      # profiling runs: the tracer's own defaults
    if args.kw:
        import json
        scheds = json.loads(args.kw)
    elif args.scan:
        scheds = [dict(spec_k=4, spec_from=int(a.split(":")[0]), spec_from2=int(a.split(":")[1])) for a in args.scan.split(",")]
    elif not args.only:
        scheds += [dict(spec_k=4, spec_k2=1), dict(spec_k=4, spec_k2=8), dict(spec_k=4, spec_from2=14), dict(spec_k=4, spec_from2=18),
                   dict(spec_k=4, spec_from=10, spec_from2=14), dict(spec_k=4, spec_from=16, spec_from2=20), dict(spec_k=4, tail_rows=2048),
                   dict(spec_k=4, tail_rows=8192), dict(spec_k=1, head_steps=args.steps, tail_rows=0), dict(spec_k=4, polish="exact")]
    for sch in [dict(dict(cone_block=(None if c < 0 else c), cone_steps=args.cone_steps), **s_) for s_ in scheds for c in args.cone]:
        tr = sdflabel_amd.SphereTracer(d, K_for(H, W), (W, H), B, steps=args.steps, device=dev, **sch)
        head, tail, spec_k, spec_from = tr.head_steps, tr.tail_rows, tr.spec_k, tr.spec_from

        def step(ev=None):
            tr.render(*prm, events=ev)
            tr.backward(g_color=o3, g_depth=o1, g_normals=o3)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = args.reps
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        evs = [{"march": (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))} for _ in range(5)]
        for e in evs:
            step(e)
        torch.cuda.synchronize()
        mm = float(np.mean([e["march"][0].elapsed_time(e["march"][1]) for e in evs]))
        s = tr.stats()
        tf = 2.0 * macs * s["ray_evaluations"] / (mm * 1e-3) / 1e12
        print("%s uniform %d cone %d polish %s spec_k %d from %2d (then %2d from %2d) head %2d tail_rows %5d: fwd+bwd %.2f ms (%.1f M rays/s), march %.2f ms, %d ray evaluations -> %.0f TFLOP/s (%.1f %% of peak), hits %d unresolved %d"
              % (str(prec).replace("torch.", ""), int(tr.uniform_tiles), tr.cone_block, tr.polish, spec_k, spec_from, tr.spec_k2, tr.spec_from2, head, tail, dt * 1e3, B * H * W / dt / 1e6, mm, s["ray_evaluations"], tf,
                 100 * tf / (2500.0 if prec == torch.float16 else 157.3), s["hits"], s["unresolved"]), flush=True)
        del tr
