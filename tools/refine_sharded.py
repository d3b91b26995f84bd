"""BASELINE configs[3] in miniature: a set of crops sharded over the GPUs of one node (one process per GPU), every rank refines its crops in
chunks with the device-resident BatchRefiner (reference losses + solver, 60 iterations, HIP-graph replay), and ONE all_gather over
RCCL brings the per-crop result rows [yaw, t(3), scale, latent(L), 2-D loss, 3-D loss] to every rank -- the only collective of the path.

    python tools/refine_sharded.py --crops 64 --chunk 16                               # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \\
           tools/refine_sharded.py --crops 1024 --chunk 64                             # 8 GPUs, 128 crops each

Synthetic crops as in bench.py: targets rendered from the ground-truth pose, perturbed initial parameters per crop.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import sdflabel_amd
from sdflabel_amd.parallel import shard_crops, gather_crop_results
from sdflabel_amd.fixtures import ASSET, K_for


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=16, help="crops refined together per launch sequence")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="float32", choices=["float32", "float16", "float32_split", "float32_prefilter"])
    ap.add_argument("--in-flight", type=int, default=1, help="chunks refined at the same time, each on a refiner and a stream of its own (r06; 2 pays "
                    "with the float16 decoder: its decoder passes run beside the other chunk's splat / loss kernels)")
    ap.add_argument("--serial-audit", action="store_true", help="the audit chain on the main stream at every batch size (for kernel time tables: "
                    "beside the candidates' pass -- the default up to 64 crops -- the overlapping kernels stretch each other's durations)")
    ap.add_argument("--reuse", action="store_true", help="candidate reuse (float16: decoder.candidate_reuse; float32_prefilter: prefilter_reuse)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    prec = {"float32": torch.float32, "float16": torch.float16}.get(args.precision, args.precision)
    dec, L = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
    dec.candidate_reuse = dec.prefilter_reuse = bool(args.reuse)
    if args.serial_audit:
        dec.candidate_audit_side_max_crops = 4
    dec = dec.to(dev)
    D, H, W = 40, args.size, args.size
    K = K_for(H, W)
    # ground truth rendering -> NOCS target and lidar cloud shared by all crops; per-crop perturbed start (seeded by the crop index)
    gt = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=dev)
    o = gt.forward(torch.tensor([0.6], device=dev), torch.tensor([[0.0, 0.0, 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
    nf = int(o["nf"][0])
    lidar = (o["xyzf"][0, :nf] * 2.0)[::2].cpu().numpy()
    target = o["color"].clone()
    from sdflabel_amd.fixtures import crop_params
    from sdflabel_amd.parallel import refine_sharded
    chunk = max(1, min(args.chunk, (args.crops + world - 1) // world))
    rfs = []
    for _ in range(max(1, args.in_flight)):
        rf = sdflabel_amd.BatchRefiner(dec, D, K, (H, W), chunk, lidar_cap=4096, device=dev)
        rf.set_crops(crop_params(list(range(chunk))), target.expand(chunk, 3, H, W), [lidar] * chunk)
        rf.capture()
        rfs.append(rf)
    rf = rfs if len(rfs) > 1 else rfs[0]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = refine_sharded(rf, crop_params(list(range(args.crops))), target, lidar, args.iters, rank, world)     # the one collective inside
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        yaw_err = (table[:, 0] - 0.6).abs()
        print("refined %d crops on %d GPU(s) in %.2f s: %.2f crops/s (%d iterations each, %s decoder); |yaw - gt| mean %.4f max %.4f; "
              "table %s" % (args.crops, world, dt, args.crops / dt, args.iters, args.precision, float(yaw_err.mean()), float(yaw_err.max()),
                            tuple(table.shape)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
