"""bench.py -- rendered rays/sec (fwd+bwd) of the differentiable SDF renderer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env; started
   as a plain `python bench.py --gpus N` the script re-launches ITSELF as N ranks (sdflabel_amd/launch.py) and refuses to run if the node has
   fewer than N GPUs -- an N-GPU line is never produced by fewer than N ranks; `rccl_ranks` in the line is dist.get_world_size())

Workload (BASELINE.json configs[1]): ONE 256x256 crop per rank, DeepSDF 8x512 decoder (L=3, latent_in=[4], weight-norm;
the committed synthetic fixture), grid density 40 (G = 64 000), float32.  One step = one refinement crop-iteration of the
reference's loop (pipelines/optimizer.py:79-123, 156) without the 2-D/3-D losses:
    decoder on the grid -> band selection -> band Jacobian (normals + d sdf/d latent) -> iso-projection -> DCM projection ->
    surfel splat + depth-softmax composite (NOCS colour, mask, normals) -> full backward to yaw, trans AND latent,
run by sdflabel_amd.BatchRenderer (B = 1 crop per rank): the same kernels as the drop-in modules, launched back to back on one
stream with device-side counts instead of host syncs; the optimizer's parameters (yaw, trans, latent) are the inputs and their
gradients the outputs.  The same crop-iteration through the drop-in Python boundary (sdflabel_amd.Grid3D / Rasterer / Decoder called
exactly as pipelines/optimizer.py calls the reference, per-iteration host syncs included) is timed too and reported as
`dropin_api`.  Nothing is cached across steps: the decoder is re-evaluated on the whole grid every step.  "march steps" in
BASELINE.json do not exist in the reference algorithm and are reported as null.
One ray = one pixel of one crop in one step; value = rays of all ranks / max-over-ranks wall time (weak scaling: one crop per
rank, no data-path collective; the per-crop results are all-gathered once after the timed region).

The printed line (ONE line, < 4 KB, key set pinned by tests/test_host_cpu.py) carries the contract keys: `roofline` for the dominant kernel (the
fused decoder forward, MFMA-bound) timed with events on the launch stream in a separate pass; `cpu_baseline`: the reference's dense algorithm as a
multi-threaded torch-CPU port (oracle/torch_cpu_port.py, pinned to the reference's golden G7) timed on the host cores for one full crop-iteration
of the same workload (rank 0, N=1 only); `refine`: crops/s of BASELINE configs[3] -- `--total-crops` (default 1024) crops sharded crop i -> rank
i mod N, refined for 60 iterations (configs/config_refine.ini:15) in chunks of 64 by BatchRefiner with the reference's losses and solver, one
all_gather of the result rows: the strong-scaling figure of north_star (seconds at 1 rank / seconds at N ranks) -- in exact f32 and in the
reference's shipped float16, both with candidate reuse (bit-identical to evaluating every grid row every iteration: DESIGN.md 3.2).
Everything else goes to bench_extras.json: the same sections evaluating every grid row (`refine_sharded_full_grid`, `refine_sharded_float16_full_grid`),
`refine_sharded_prefilter`, `refine_sharded_configs4` (512x512 rays, float16), `refine_sharded_traced`, `roofline_splat`, `dropin_api`, `pose_only`,
`f16_decoder`, `split_decoder`, `prefilter_decoder`, `sphere_trace`, `optimizer_mirror_varied_crops`, per-rank step times.  Under --gpus N > 1 only
the headline and the two `refine` sections run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
D, H, W = 40, 256, 256


from sdflabel_amd.fixtures import ASSET, K_for, crop_params, crop_start, fitted_state, kitti_like_problems, synthetic_targets  # noqa: E402


def build_pose(yaw, trans):
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = yaw.new_zeros(1), yaw.new_ones(1)
    pose = torch.eye(4, device=yaw.device)
    pose[:3, :3] = torch.stack((c, z, s, z, o, z, -s, z, c)).view(3, 3)   # utils/refinement.py:108-125
    pose[1] *= -1                                                          # optimizer.py:88-90
    pose[:3, 3] = trans
    return pose


class Crop:
    """One synthetic refinement problem: GT pose yaw .6, t (0,0,3.5); init perturbed per crop index."""

    def __init__(self, index, dev):
        yaw, trans, latent = crop_start(index)
        self.yaw = torch.from_numpy(yaw).to(dev).requires_grad_(True)
        self.trans = torch.from_numpy(trans).to(dev).requires_grad_(True)
        self.latent = torch.from_numpy(latent).to(dev).requires_grad_(True)


def crop_iteration(dec, grid, renderer, crop, ev=None):
    for p in (crop.yaw, crop.trans, crop.latent):
        p.grad = None
    latent_ = F.normalize(crop.latent, p=2, dim=0)                                         # optimizer.py:96
    inputs = torch.cat([latent_.expand(grid.points.size(0), -1), grid.points], 1)          # :99-100
    if ev is not None:
        ev[0].record()
    sdf, _ = dec(inputs)                                                                   # :101
    if ev is not None:
        ev[1].record()
    pcd, _, normals = grid.get_surface_points(sdf)                                         # :104
    pose = build_pose(crop.yaw, crop.trans)
    rendering, points = renderer(pcd, normals, normals, pose, primitives='disc', rot='dcm', bg=None, output_depth=False,
                                 output_normals=True, output_nocs=True, output_points=True, output_mask=True)   # :110-123
    loss = rendering['color'].sum() + rendering['mask'].sum() + rendering['normals'].sum() + points['xyzf'].sum()
    loss.backward()                                                                        # :156
    return loss.detach(), pcd.shape[0], points['xyzf'].shape[0]


def cpu_baseline():
    """The reference's dense algorithm on the host cores: ONE full crop-iteration (fwd+bwd) of the bench workload -- 256x256 rays, D = 40,
    every pixel, nothing extrapolated -- with oracle/torch_cpu_port.py (torch CPU ops in the reference's order, autograd backward incl. the
    decoder's unneeded weight gradients, dense N x P splat tensors; measured in the build container at the cost of the imported reference
    itself: 14.4 s against 13.8 s on 8 cores).  Timed with 8 threads (the survey's probe configuration) and with 32."""
    from oracle import torch_cpu_port as TP
    st, spec = fitted_state()
    decoder = TP.DecoderPort(st, spec)
    gp = TP.generate_point_grid(D).requires_grad_(True)
    K = torch.from_numpy(K_for(H, W))
    ncpu = os.cpu_count() or 1
    prev = torch.get_num_threads()
    out, n_surf = {}, 0

    def run():
        # exactly the GPU step's input: synthetic crop 0's perturbed start (the same surfel count N as the timed GPU step)
        y0, t0_, l0 = crop_start(0)
        yaw = torch.from_numpy(y0.copy()).requires_grad_(True)
        trans = torch.from_numpy(t0_.copy()).requires_grad_(True)
        lat = torch.from_numpy(l0.copy()).requires_grad_(True)
        t0 = time.perf_counter()
        rend, pts, n, loss = TP.crop_iteration(decoder, gp, K, (W, H), yaw, trans, lat)
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(yaw.grad).all() and torch.isfinite(trans.grad).all() and torch.isfinite(lat.grad).all())
        return dt, n

    # 8 threads = the survey's probe configuration (median of 3 timed iterations after one warm-up); 32 = a quarter of a socket, one timed
    # iteration beside it.  (Every hardware thread of the 256-thread GPU box measured 1.08 k rays/s, 60 s per iteration: the dense passes are
    # memory-bound and oversubscribe -- not timed by default)
    t8, t32 = min(8, ncpu), min(32, ncpu)
    torch.set_num_threads(t8)
    run()                                           # warm-up (allocator, thread pool)
    samples = []
    for _ in range(3):
        dt, n_surf = run()
        samples.append(dt)
    med = float(np.median(samples))
    out[t8] = (H * W / med, med)
    if t32 != t8:
        torch.set_num_threads(t32)
        dt, n_surf = run()
        out[t32] = (H * W / dt, dt)
    torch.set_num_threads(prev)
    return {"value": out[t8][0], "unit": "rays/s", "cores": int(t8), "kind": "port",
            "seconds_per_crop_iteration_samples": samples,
            "by_threads": {str(k): {"rays_per_s": v[0], "seconds_per_crop_iteration": v[1]} for k, v in out.items()},
            "host_cores": ncpu,
            "sample": "one full crop-iteration (fwd+bwd) of the bench workload: all %dx%d rays, D=%d, N=%d surfels, the reference's dense N x P "
                      "algorithm as torch-CPU ops (oracle/torch_cpu_port.py); median of 3 on %d threads (+1 on %d threads in the extras)" % (H, W, D, n_surf, t8, t32)}


# ---- the printed line -------------------------------------------------------------------------------------------------------------------
# The driver parses the LAST stdout line.  It carries the contract keys and nothing else; every other section of a run lives in
# bench_extras.json.  tests/test_host_cpu.py pins the key set and the 4 KB bound.
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline", "refine", "extras")
CONFIG_KEYS = ("workload", "crops_per_gpu", "rays_per_crop", "grid_points", "surfels", "front_facing", "march_steps", "parallelism")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "flops_per_launch", "avg_launch_ms")
CPU_KEYS = ("value", "unit", "cores", "kind", "host_cores", "sample")
LINE_MAX_BYTES = 4096


# a full-size record (the sections of the r04 run, abridged) for the line tests and --line-check
CANNED_LINE = {
    "metric": "rendered rays/sec (fwd+bwd)", "value": 34243950.123456, "unit": "rays/s", "n_gpus": 1, "rccl_ranks": 1, "steps": 250, "warmup": 10,
    "ms_per_step": 1.9137984, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    "config": {"workload": "BASELINE configs[1]: one 256x256 crop per GPU, DeepSDF 8x512 on a 40^3 grid, fwd+bwd to yaw/trans/latent", "crops_per_gpu": 1,
               "rays_per_crop": 65536, "grid_points": 64000, "surfels": 2751, "front_facing": 763, "march_steps": None, "parallelism": "crop-parallel x1"},
    "roofline": {"kernel": "sdfr_mlp_kernel<float,32,2,2,8,2,1,2> (fused decoder forward)", "bound": "mfma", "achieved": 137.0297, "peak": 157.3,
                 "unit": "TFLOP/s", "frac": 0.871136, "traffic": 271958400.0, "flops_per_launch": 2.349466e11, "avg_launch_ms": 1.714567,
                 "timing": "x" * 300, "traffic_source": "y" * 300},
    "cpu_baseline": {"value": 6137.024, "unit": "rays/s", "cores": 8, "kind": "port", "host_cores": 256, "sample": "z" * 900,
                     "by_threads": {"8": {}, "32": {}}, "seconds_per_crop_iteration_samples": [10.6, 10.7, 10.8]},
    "refine_sharded": {"crops_per_s": 9.508298, "total_crops": 1024, "iterations_per_crop": 60, "workload": "w" * 400},
    "refine_sharded_float16": {"crops_per_s": 71.89664, "total_crops": 1024, "iterations_per_crop": 60, "candidate_reuse": True,
                               "full_grid_passes_per_crop_last_chunk_mean": 1.0},
    "refine_sharded_area32": {"crops_per_s": 301.5, "total_crops": 1024, "workload": "a" * 400},
    "sphere_trace": {"f16_64_steps": {"note": "n" * 5000}}, "dropin_api": {"launches": {"k": list(range(500))}}, "per_rank_ms_per_step": [1.9] * 8,
}


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def _num(v):
    # 7 significant digits are plenty for a printed rate; keeps the line small
    return float("%.7g" % v) if isinstance(v, float) else v


def compact_line(full, extras_path=None):
    """The one JSON line bench.py prints: the contract keys of `full` (the dict with every section of the run), strings bounded, < 4 KB."""
    out = {}
    for k in LINE_KEYS:
        if k in ("config", "roofline", "cpu_baseline", "refine", "extras"):
            continue
        out[k] = _num(full.get(k))
    cfg = full.get("config") or {}
    out["config"] = {k: _short(cfg.get(k), 200) for k in CONFIG_KEYS if k in cfg}
    rf = full.get("roofline") or {}
    out["roofline"] = {k: _num(_short(rf.get(k), 120)) for k in ROOFLINE_KEYS if k in rf}
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = {k: _num(_short(cb.get(k), 240)) for k in CPU_KEYS if k in cb} if isinstance(cb, dict) else None
    # the metric's second half ("refine-demo crops/sec"): the sharded 60-iteration refinement, exact f32 and the reference's shipped f16
    ref = {}
    for key, name in (("refine_sharded", "f32"), ("refine_sharded_float16", "f16")):
        sec = full.get(key)
        if isinstance(sec, dict) and "crops_per_s" in sec:
            ref[name + "_crops_per_s"] = _num(float(sec["crops_per_s"]))
            ref["total_crops"] = sec.get("total_crops")
            ref["iterations_per_crop"] = sec.get("iterations_per_crop")
            if "candidate_reuse" in sec:                     # (bit-identical to evaluating every grid row every iteration; the full-grid figures are in the extras)
                ref[name + "_candidate_reuse"] = bool(sec["candidate_reuse"])
            if "full_grid_passes_per_crop_last_chunk_mean" in sec:      # (of iterations_per_crop decoder steps, how many ran the whole grid)
                ref[name + "_full_grid_passes_per_crop"] = _num(float(sec["full_grid_passes_per_crop_last_chunk_mean"]))
    a32 = full.get("refine_sharded_area32")
    if isinstance(a32, dict) and "crops_per_s" in a32:       # the reference's shipped operating point (rendering_area 32, float16), batched
        ref["area32_f16_crops_per_s"] = _num(float(a32["crops_per_s"]))
    out["refine"] = ref or None
    out["extras"] = extras_path
    line = json.dumps(out)
    if len(line.encode()) >= LINE_MAX_BYTES:
        raise RuntimeError("bench line of %d bytes: the driver's parser needs < %d" % (len(line.encode()), LINE_MAX_BYTES))
    return line


def write_extras(full, path=None):
    """Every section of the run as indented JSON next to bench.py (and under gpurun_out/ when that directory exists: it is what a gpurun call
    merges back).  Returns the repo-relative path written, or None when the directory is read-only (the line is printed regardless)."""
    path = path or os.path.join(ROOT, "bench_extras.json")
    written = None
    for p in (path, os.path.join(ROOT, "gpurun_out", "bench_extras.json")):
        if p != path and not os.path.isdir(os.path.dirname(p)):
            continue
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extras", default=None, help="where the full record of the run is written (default: bench_extras.json next to bench.py)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--crops-per-gpu", type=int, default=1, help="crops refined together per rank (1 = BASELINE configs[1]; 64 = configs[2])")
    ap.add_argument("--crop-size", type=int, default=256, help="crop edge in pixels (256 = BASELINE configs[1..3]; 512 = configs[4], informational)")
    ap.add_argument("--total-crops", type=int, default=1024, help="crops of the sharded refinement (BASELINE configs[3]); 0 skips the sections")
    ap.add_argument("--sharded-iters", type=int, default=60, help="refinement iterations per crop in the sharded sections (the reference's "
                    "refinement length, configs/config_refine.ini:15)")
    ap.add_argument("--configs4-crops", type=int, default=256, help="crops of the configs[4]-shaped sharded section (512x512 rays, float16 decoder)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline loop (+ cpu_baseline): skip the informational sections")
    ap.add_argument("--line-check", action="store_true", help="with --launch-check: print the compact bench line (canned sections, no "
                    "measurement) instead of the launch report")
    ap.add_argument("--launch-check", action="store_true", help="only bring up the N ranks, all_gather their ranks and print one JSON line "
                    "(plumbing check of the self-launcher; falls back to gloo on a machine without GPUs)")
    args = ap.parse_args()
    global H, W
    H = W = int(args.crop_size)

    from sdflabel_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # plain `python bench.py --gpus N`: become N ranks (never fall through to a one-GPU run that prints an N-GPU line)
        os.environ["SDFR_SELF_LAUNCHED"] = "1"
        rc = launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                require_gpus=not (args.launch_check and not torch.cuda.is_available()))
        raise SystemExit(rc)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: the launcher's rank count and --gpus must agree" % (args.gpus, world))
    launcher = "self (sdflabel_amd/launch.py)" if os.environ.get("SDFR_SELF_LAUNCHED") == "1" else (
        "torch.distributed.run (external)" if launch.under_launcher() else "none (single process)")
    has_gpu = torch.cuda.is_available()
    if not has_gpu and not args.launch_check:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if has_gpu:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d has no GPU (device_count %d)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if has_gpu else torch.device("cpu")
    if world > 1:                                              # each rank its share of the host cores
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    dist = None
    backend = None
    if world > 1 or launch.under_launcher():       # under a launcher: always take the distributed path
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if has_gpu else "gloo"
        if has_gpu:
            dist_mod.init_process_group(backend, device_id=dev)
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
        if dist.get_world_size() != world:
            raise SystemExit("process group of %d ranks, WORLD_SIZE %d" % (dist.get_world_size(), world))
    rccl_ranks = dist.get_world_size() if dist is not None else 1
    rccl_version = None
    if has_gpu and dist is not None:
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_version = None
    if args.launch_check:
        ranks = [rank]
        if dist is not None:
            t = torch.tensor([rank], dtype=torch.int64, device=dev)
            got = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(got, t)
            ranks = [int(g.item()) for g in got]
        if rank == 0 and args.line_check:
            # the compact line as a real run assembles it, with no measurement in it (value null): what the gloo test parses
            print(compact_line(dict(CANNED_LINE, n_gpus=world, rccl_ranks=rccl_ranks, value=None, ms_per_step=None, data="none (--line-check)"),
                               None), flush=True)
        elif rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": rccl_ranks, "backend": backend, "rccl_version": rccl_version,
                              "launcher": launcher, "ranks": ranks, "devices": torch.cuda.device_count() if has_gpu else 0}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import sdflabel_amd
    if not os.path.isfile(sdflabel_amd.LIB_PATH):             # fresh checkout on the GPU box: compile the HIP library once (rank 0)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    dec = dec.to(dev)
    CB = args.crops_per_gpu
    from sdflabel_amd.parallel import gather_crop_results, refine_sharded, shard_crops
    crops = [Crop(i, dev) for i in shard_crops(CB * world, rank, world)]
    crop = crops[0]
    macs = dec.handle(dev).macs
    br = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), CB, device=dev)
    G = br.G
    br.set_params(torch.cat([c.yaw.detach() for c in crops]), torch.stack([c.trans.detach() for c in crops]),
                  torch.stack([c.latent.detach() for c in crops]))
    ones3 = torch.ones(CB, 3, H, W, device=dev)
    ones1 = torch.ones(CB, 1, H, W, device=dev)
    onesx = torch.ones(CB, br.cap, 3, device=dev)

    def step(ev=None, evs=None):
        br.forward(mlp_events=ev, events=evs)
        br.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx, events=evs)     # d/d(out) of the plain sums used as the loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    # informational sections run in a one-rank job only: under --gpus N > 1 the command is the headline + the configs[3] sharded refinement in
    # f32 and f16 (the strong-scaling figure), so that an 8-rank run stays well under 300 s
    extras = (not args.no_extras) and world == 1

    def ev_pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    import gc
    # ---- the headline: EXACTLY --steps steps, nothing but the launches inside the timed region (no event records, no collector pause) ----
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local)
    gc.enable()
    per_rank_ms = [dt_local / args.steps * 1e3]
    if dist is not None:                       # every rank's own step time (extras file only; the line's ms_per_step is their maximum)
        t_ = torch.tensor([per_rank_ms[0]], device=dev, dtype=torch.float64)
        got = [torch.zeros_like(t_) for _ in range(world)]
        dist.all_gather(got, t_)
        per_rank_ms = [float(g.item()) for g in got]
    assert not br.overflow()
    n_surf, n_front = int(br.cnt[0]), int(br.fcnt[0])
    loss = br.color[0].sum() + br.mask[0].sum() + br.nimg[0].sum() + br.xyzf[0].sum()
    if dist is not None:
        # the path's only exchange: per-crop result rows gathered once, outside the per-iteration critical path
        res = torch.cat([br.color.sum(dim=(1, 2, 3)).view(CB, 1), br.g_yaw.view(CB, 1), br.g_trans, br.g_latent], dim=1).float()
        table = gather_crop_results(res, CB * world, rank, world)
        assert table.shape == (CB * world, 8) and bool(torch.isfinite(table).all())
    # ---- a longer run of the same step (>= 200 steps and >= 0.5 s): the headline's 20-250 steps are a 40-500 ms region ----
    long_steps = max(200, int(0.6 / max(dt / args.steps, 1e-6)) + 1)
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(long_steps):
        step()
    barrier()
    dt_long = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    # ---- kernel durations: a SEPARATE pass with event pairs around the dominant launches, on the launch stream ----
    ev_steps = max(20, min(args.steps, 100))
    events = [ev_pair() for _ in range(ev_steps)]
    kev = [{"jacobian": ev_pair(), "splat_fwd": ev_pair(), "splat_bwd": ev_pair()} for _ in range(ev_steps)]
    for i in range(ev_steps):
        step(events[i], kev[i])
    torch.cuda.synchronize()
    mlp_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    kms = {k: float(np.mean([e[k][0].elapsed_time(e[k][1]) for e in kev])) for k in kev[0]}

    def all_ok(flag):
        if dist is None:
            return flag
        t_ = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.MIN)
        return bool(t_.item())

    def timed_section(setup, run):
        """Informational measurement that cannot deadlock a multi-rank run: whatever fails locally, every rank executes the same
        sequence of collectives.  Returns ((state, seconds), None) or (None, error string)."""
        state, err = None, None
        try:
            state = setup()
        except Exception as e:
            err = repr(e)[:200]
        if not all_ok(err is None):
            return None, err or "failed on another rank"
        gc.collect()
        gc.disable()
        barrier()
        t_ = time.perf_counter()
        try:
            run(state)
        except Exception as e:
            err = repr(e)[:200]
        barrier()
        d_ = max_over_ranks(time.perf_counter() - t_)
        gc.enable()
        if not all_ok(err is None):
            return None, err or "failed on another rank"
        return (state, d_), None

    # the full refinement loop (reference: 60 iterations per crop, configs/config_refine.ini:15) with the reference's 2-D and 3-D losses and
    # its Adam/SGD step, device resident (sdflabel_amd.BatchRefiner); targets are rendered from the ground-truth pose (SURVEY.md 8 a-harness)
    iters = 60

    def refine_setup(reuse=False):
        rf = sdflabel_amd.BatchRefiner(dec, D, K_for(H, W), (H, W), CB, lidar_cap=4096, device=dev, candidate_reuse=reuse)
        nocs1, lidar = synthetic_targets(dec, D, K_for(H, W), H, W, dev)
        nocs_t = nocs1.expand(CB, 3, H, W).clone()
        p0 = {"yaw": torch.cat([c.yaw.detach() for c in crops]), "trans": torch.stack([c.trans.detach() for c in crops]),
              "scale": torch.full((CB,), 2.0), "latent": torch.stack([c.latent.detach() for c in crops])}
        rf.set_crops(p0, nocs_t, [lidar] * CB)
        rf.capture()
        rf.optimize(3)                                       # warm-up
        rf.set_crops(p0, nocs_t, [lidar] * CB)               # restart from the initial parameters
        return rf, p0["yaw"].to(dev).clone()

    res, err = timed_section(refine_setup, lambda st: st[0].optimize(iters)) if extras else (None, "skipped (--no-extras or world > 1)")
    if res is None:
        refine = {"error": err}
    else:
        (rf, y0), dt_r = res
        refine = {"value": CB * world / dt_r, "unit": "crops/s", "iterations_per_crop": iters, "ms_per_iteration": dt_r / iters * 1e3,
                  "crops": CB * world, "losses": "reference 2-D NOCS window loss + 3-D nearest-neighbour loss, Adam/SGD step, on device",
                  "yaw_error_before_after": [float((y0 - 0.6).abs().mean()), float((rf.yaw - 0.6).abs().mean())],
                  "crops_stepped_last_iteration": int(rf.stepped.sum())}
        del rf
        # ... and with candidate reuse (exact f32, bit-identical: DESIGN.md 3.2)
        res2, err2 = timed_section(lambda: refine_setup(True), lambda st: st[0].optimize(iters))
        if res2 is not None:
            (rf2, _), dt_r2 = res2
            refine["with_candidate_reuse"] = {"value": CB * world / dt_r2, "unit": "crops/s", "ms_per_iteration": dt_r2 / iters * 1e3,
                                              "yaw_error_after": float((rf2.yaw - 0.6).abs().mean())}
            del rf2
        else:
            refine["with_candidate_reuse"] = {"error": err2}
    res = None

    # the same loop with the sphere tracer as its renderer (BatchRefiner(render="trace"): traced NOCS image -> 2-D loss, hit points -> 3-D loss,
    # surfel-semantics backward -> the same solver step; float16 decoder = the reference's shipped precision; NOT the reference's algorithm)
    def traced_setup():
        d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
        rf = sdflabel_amd.BatchRefiner(d16.to(dev), D, K_for(H, W), (H, W), CB, lidar_cap=4096, device=dev, render="trace")
        nocs1, lidar = synthetic_targets(dec, D, K_for(H, W), H, W, dev)
        nocs_t = nocs1.expand(CB, 3, H, W).clone()
        p0 = {"yaw": torch.cat([c.yaw.detach() for c in crops]), "trans": torch.stack([c.trans.detach() for c in crops]),
              "scale": torch.full((CB,), 2.0), "latent": torch.stack([c.latent.detach() for c in crops])}
        rf.set_crops(p0, nocs_t, [lidar] * CB)
        rf.capture()
        rf.optimize(3)
        rf.set_crops(p0, nocs_t, [lidar] * CB)
        return rf, p0["yaw"].to(dev).clone(), p0["trans"].to(dev).clone()

    res, err = timed_section(traced_setup, lambda st: st[0].optimize(iters)) if extras else (None, "skipped (--no-extras or world > 1)")
    if res is None:
        refine_traced = {"error": err}
    else:
        (rf, y0, t0v), dt_r = res
        gt_t = torch.tensor([0.0, 0.0, 3.5], device=dev)
        refine_traced = {"value": CB * world / dt_r, "unit": "crops/s", "iterations_per_crop": iters, "ms_per_iteration": dt_r / iters * 1e3,
                         "crops": CB * world, "renderer": "sphere tracer (cone marching on %dx%d-pixel tiles, speculative passes), float16 decoder, "
                         "surfel-semantics backward; reference losses + solver on device; HIP-graph replay" % (rf.tr.cone_block, rf.tr.cone_block),
                         "rays_per_s_incl_losses_and_solver": CB * world * H * W * iters / dt_r,
                         "yaw_error_before_after": [float((y0 - 0.6).abs().mean()), float((rf.yaw - 0.6).abs().mean())],
                         "trans_error_before_after": [float((t0v - gt_t).abs().max(1)[0].mean()), float((rf.trans - gt_t).abs().max(1)[0].mean())],
                         "hits_per_crop_last_iteration": float(rf.tr.ecnt.float().mean()),
                         "crops_stepped_last_iteration": int(rf.stepped.sum())}
        del rf
    res = None

    # ---- the product-side Optimizer on crops as the reference PIPELINE produces them: every annotation its own crop size and
    # intrinsics (utils/refinement.py:586-609 adjust_intrinsics_crop: area-normalised to rendering_area^2, aspect of the 2-D box kept, principal
    # point moved by the box corner; pipelines/refine_css.py:203-223 constructs an Optimizer per annotation).  32 KITTI-like boxes, one
    # Optimizer(...).optimize(60, ...) call per crop as the pipeline issues them; the time INCLUDES building the refiner (buffers + HIP-graph
    # capture), which ragged extents make a once-per-capacity cost: refiners_built / graph_captures should read 1 / 1 per rendering area.
    d16_shared = []

    def varied_crops(area, render="splat"):
        from sdflabel_amd.pipelines import optimizer as OP
        OP.clear_refiner_cache()
        OP.STATS["refiners_built"] = 0
        if not d16_shared:                 # ONE decoder object for the three sections, as a pipeline has (its Lipschitz bound is cached on it)
            d16_, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
            d16_shared.append(d16_.to(dev))
            t_l = time.perf_counter()
            d16_shared[0].latent_lipschitz_bound()
            d16_shared.append(time.perf_counter() - t_l)
        d16 = d16_shared[0]
        rng = np.random.default_rng(11)
        n = 32
        boxes_w = rng.uniform(60, 420, n)
        boxes_h = boxes_w / rng.uniform(1.2, 3.2, n)                                         # cars: 1.2 ... 3.2 times as wide as high
        shapes, Ks, gts = [], [], []
        for bw, bh in zip(boxes_w, boxes_h):
            r = np.sqrt(area * area / (bh * bw))
            Hc, Wc = int(bh * r), int(bw * r)                                                 # crop_size.int() (:603)
            f = 1.15 * Hc * 3.5 / 2.0                                                         # the object (a 2-unit cube at z = 3.5) about fills the crop's height
            cx, cy = rng.uniform(-1.0 * Wc, 2.0 * Wc), rng.uniform(0.2 * Hc, 0.8 * Hc)        # principal point far outside the crop, as after the box-corner shift
            shapes.append((Hc, Wc))
            Ks.append(np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], np.float32))
            gts.append(np.array([3.5 * (Wc / 2.0 - cx) / f, 3.5 * (Hc / 2.0 - cy) / f, 3.5], np.float32))
        pmax = 1 << (max(h * w for h, w in shapes) - 1).bit_length()
        # targets: the ground-truth pose of every crop rendered in ONE ragged batch (exact-f32 decoder)
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
        t_ = time.perf_counter()
        errs0, errs1, caps = [], [], set()
        for b in range(n):
            y0, t0_, l0 = starts[b]
            p = {"yaw": y0.copy(), "trans": (gts[b] + (t0_ - np.asarray([0.0, 0.0, 3.5], np.float32))).astype(np.float32), "scale": np.array([2.0], np.float32),
                 "latent": l0.copy()}
            opt = OP.Optimizer(p, dev, {"2d": 0.3, "3d": 0.5}, render=render)
            out = opt.optimize(iters, targets[b], lidars[b], d16, grid, torch.from_numpy(Ks[b]), list(shapes[b]))
            caps.add(id(opt._refiner))
            errs0.append(abs(float(y0[0]) - 0.6)); errs1.append(abs(float(out["yaw"][0]) - 0.6))
        torch.cuda.synchronize()
        dt_v = time.perf_counter() - t_
        captures = sum(getattr(v[1], "captures", 0) for v in OP._REFINERS.values())
        res_ = {"value": n / dt_v, "unit": "crops/s", "crops": n, "iterations_per_crop": iters, "rendering_area": area, "seconds_incl_refiner_construction": dt_v,
                "crop_sizes_h_w_min_max": [list(min(shapes)), list(max(shapes))], "distinct_crop_sizes": len(set(shapes)), "pixel_capacity": pmax,
                "refiners_built": OP.STATS["refiners_built"], "graph_captures": captures, "distinct_refiners_used": len(caps),
                "decoder_lipschitz_bound_seconds_once_per_decoder_not_included": d16_shared[1],
                "renderer": "surfel splat (the reference's algorithm)" if render == "splat" else "sphere tracer (Optimizer(..., render='trace'))",
                "decoder_precision": "float16 (the reference's shipped precision)", "mean_abs_yaw_error_before_after": [float(np.mean(errs0)), float(np.mean(errs1))],
                "call": "Optimizer(params, device, weights).optimize(60, nocs, lidar, dsdf, grid, K_b, [H_b, W_b]) per crop, as pipelines/refine_css.py:203-223"}
        OP.clear_refiner_cache()
        return res_

    varied = None
    if rank == 0 and CB == 1 and extras:
        varied = {}
        for area, render in ((32, "splat"), (256, "splat"), (256, "trace")):
            key = "rendering_area_%d" % area + ("" if render == "splat" else "_traced")
            try:
                varied[key] = varied_crops(area, render)
            except Exception as e:
                varied[key] = {"error": repr(e)[:300]}

    # ---- BASELINE configs[3]: `--total-crops` crops sharded over the ranks (crop i -> rank i mod N), refined for the reference's 60 iterations
    # in chunks of 64 by BatchRefiner, ONE all_gather of the per-crop result rows at the end (sdflabel_amd.parallel.refine_sharded).
    # Strong scaling: the total is fixed, so seconds(N=1) / seconds(N) is the north_star's "x at 8 GPUs over 1 GPU on a 1024-crop batch".
    # Three decoder arithmetics, labelled: exact float32 (the parity path, headline), float16 (the reference's shipped precision,
    # config_refine.ini:19; pinned to the reference's own float16 trajectory, golden G8h) and float32_prefilter + candidate reuse (exact
    # float32 on everything consumed downstream; guarded at run time); and the configs[4] shape (512x512 rays, float16 decoder).
    def phase_table(tm):
        """every rank's seconds in set_crops / optimize (enqueue + GPU, synchronised per chunk) / all_gather of a sharded section, gathered on all
        ranks (r06: under world > 1 the driver's curve can be read -- which phase stops scaling)"""
        keys = ("set_crops", "optimize", "all_gather", "chunks")
        mine = torch.tensor([float(tm.get(k, 0.0)) for k in keys], dtype=torch.float64, device=dev)
        rows = [mine]
        if dist is not None:
            rows = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
        return {"rank_%d" % r: {k: float(v) for k, v in zip(keys, row.tolist())} for r, row in enumerate(rows)}

    def sharded_section(label, precision, reuse, size, total, workload, render="splat", in_flight=1):
        chunk = max(1, min(64, (total + world - 1) // world))
        Kc = K_for(size, size)
        # in_flight (r06): chunks refined at the same time, each on a refiner and a stream of its own (sdflabel_amd.parallel.refine_sharded with a
        # list of refiners): the matrix-core-bound decoder passes of one chunk run beside the VALU-bound splat / loss kernels of the other.  Pays
        # with the float16 decoder (+10 %); the exact-f32 iteration is 95 % decoder passes: one refiner
        in_flight = max(1, min(in_flight, (((total + world - 1) // world) + chunk - 1) // chunk))

        def setup():
            d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
            d2.prefilter_reuse = reuse
            d2.candidate_reuse = reuse                         # (float16 / exact float32: candidate rows only while the proven bound holds, r05)
            d2 = d2.to(dev)
            nocs1, lidar = synthetic_targets(dec, D, Kc, size, size, dev)      # targets from the exact-f32 rendering of the ground truth, all modes
            rfs = []
            for _ in range(in_flight):
                rf = sdflabel_amd.BatchRefiner(d2, D, Kc, (size, size), chunk, lidar_cap=4096, device=dev, render=render)
                rf.set_crops(crop_params(list(range(chunk))), nocs1.expand(chunk, 3, size, size), [lidar] * chunk)
                rf.capture()
                rf.optimize(2)                                 # warm-up (graph instantiation, allocator)
                rfs.append(rf)
            return [rfs, crop_params(list(range(total))), nocs1, lidar]

        def run(st):
            rfs, params, nocs1, lidar = st[:4]
            tm = {}
            st.append(refine_sharded(rfs if len(rfs) > 1 else rfs[0], params, nocs1, lidar, args.sharded_iters, rank, world, timing=tm))
            st.append(tm)

        res_, err_ = timed_section(setup, run)
        if res_ is None:
            return {"label": label, "error": err_}
        st, dt_s = res_
        rf, table, tm = st[0][0], st[-2], st[-1]
        ok = tuple(table.shape) == (total, 7 + rf.L) and bool(torch.isfinite(table).all())
        p_all = st[1]
        out = {"label": label, "workload": workload % (total, size, size, world, chunk), "decoder_precision": str(precision).replace("torch.", ""),
               "total_crops": total, "iterations_per_crop": args.sharded_iters, "world_size": world, "rccl_ranks": rccl_ranks, "seconds": dt_s,
               "crops_per_s": total / dt_s, "crop_iterations_per_s": total * args.sharded_iters / dt_s,
               "rays_per_s_incl_losses_and_solver": total * args.sharded_iters * size * size / dt_s,
               "mean_abs_yaw_error_before_after": [float(np.abs(p_all["yaw"] - 0.6).mean()), float((table[:, 0] - 0.6).abs().mean())],
               "gathered_row": "yaw, trans(3), scale, latent(%d), weighted 2-D loss, weighted 3-D loss" % rf.L,
               "mean_weighted_losses_2d_3d_after": [float(table[:, -2].mean()), float(table[:, -1].mean())],
               "gathered_table_ok": ok, "scaling": "strong (total crops fixed): speed-up at N ranks = seconds(N=1) / seconds(N)",
               "chunks_in_flight_per_rank": in_flight, "per_rank_phase_seconds": phase_table(tm)}
        if render != "splat":
            out["renderer"] = "sphere tracer (BatchRefiner(render='trace'), surfel-semantics backward): not the reference's algorithm"
        if rf.br is not None and getattr(rf.br, "guarded", False):
            out["guard"] = rf.br.prefilter_report()
            out["candidate_reuse"] = bool(rf.br.reuse)
            out["lipschitz_bound_latent"] = rf.br.lipschitz
            out["margin"] = rf.br.margin
            out["candidates_per_crop_last_chunk_mean_max"] = [float(rf.br.ccnt.float().mean()), int(rf.br.ccnt.max())]
            out["full_grid_passes_per_crop_last_chunk_mean"] = float(rf.br.n_full.float().mean())
        del st, rf
        return out

    # ---- the reference's SHIPPED operating point, batched: configs/config_refine.ini:11-19 = grid_density 40, rendering_area 32,
    # iters 60, precision float16.  `total` KITTI-like crops (64 distinct problems of sdflabel_amd.fixtures.kitti_like_problems: every crop its own
    # (H_b, W_b) with H_b W_b ~ 32^2 and its own intrinsics, utils/refinement.py:586-609; repeated to `total` with per-crop starts), sharded crop i ->
    # rank i mod N, ragged chunks of 64 through ONE BatchRefiner(max_pixels=...) and ONE captured graph, one all_gather.
    def sharded_area32(total, area=32):
        chunk = max(1, min(64, (total + world - 1) // world))
        distinct = 64

        def setup():
            d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
            d2.candidate_reuse = True
            d2 = d2.to(dev)
            shapes, Ks, targets, lidars, starts = kitti_like_problems(dec, D, area, distinct, dev)
            pmax = max(1024, 1 << (max(h * w for h, w in shapes) - 1).bit_length())
            ids = [i % distinct for i in range(total)]
            params = {k: np.stack([starts[j][k].reshape(-1) for j in ids]) for k in ("yaw", "trans", "scale", "latent")}
            params["yaw"] = params["yaw"] + 0.01 * (np.arange(total, dtype=np.float32) // distinct).reshape(-1, 1)      # (repeats start elsewhere)
            sel = list(range(chunk))
            rfs = []
            for _ in range(2 if (total + world - 1) // world > chunk else 1):          # two chunks in flight per rank (see sharded_section)
                rf = sdflabel_amd.BatchRefiner(d2, D, Ks[0], shapes[0], chunk, lidar_cap=max(1024, 1 << (max(l.shape[0] for l in lidars) - 1).bit_length()),
                                               device=dev, max_pixels=pmax, candidate_reuse=True)
                rf.set_crops({k: v[sel] for k, v in params.items()}, [targets[ids[i]] for i in sel], [lidars[ids[i]] for i in sel],
                             K=np.stack([Ks[ids[i]] for i in sel]), crop_sizes=[shapes[ids[i]] for i in sel])
                rf.capture()
                rf.optimize(2)
                rfs.append(rf)
            return [rfs, params, [targets[j] for j in ids], [lidars[j] for j in ids], np.stack([Ks[j] for j in ids]), [shapes[j] for j in ids], shapes]

        def run(st):
            rfs, params, tg, li, Kall, sz = st[:6]
            tm = {}
            st.append(refine_sharded(rfs if len(rfs) > 1 else rfs[0], params, tg, li, args.sharded_iters, rank, world, K=Kall, crop_sizes=sz, timing=tm))
            st.append(tm)

        res_, err_ = timed_section(setup, run)
        if res_ is None:
            return {"error": err_}
        st, dt_s = res_
        rf, table, tm, shapes = st[0][0], st[-2], st[-1], st[6]
        n_flight = len(st[0])
        y0 = st[1]["yaw"].reshape(-1)
        out = {"label": "the reference's shipped operating point (configs/config_refine.ini:11-19), batched: float16 decoder, candidate reuse, ragged crops",
               "workload": "%d KITTI-like crops at rendering_area %d (own (H, W) and K per crop, %d distinct problems), 60 iterations, sharded crop i -> rank i mod %d, "
                           "ragged chunks of %d through one BatchRefiner + one captured graph per chunk in flight (%d), one all_gather" % (total, area, distinct, world, chunk, n_flight),
               "total_crops": total, "iterations_per_crop": args.sharded_iters, "world_size": world, "seconds": dt_s, "crops_per_s": total / dt_s,
               "crop_sizes_h_w_min_max": [list(min(shapes)), list(max(shapes))], "rendering_area": area, "graph_captures": getattr(rf, "captures", None),
               "mean_abs_yaw_error_before_after": [float(np.abs(y0 - 0.6).mean()), float((table[:, 0] - 0.6).abs().mean())],
               "gathered_table_ok": tuple(table.shape) == (total, 7 + rf.L) and bool(torch.isfinite(table).all()),
               "candidate_reuse": bool(rf.br.creuse), "full_grid_passes_per_crop_last_chunk_mean": float(rf.br.n_full.float().mean()),
               "guard": rf.br.prefilter_report(), "per_rank_phase_seconds": phase_table(tm)}
        del st, rf
        return out

    sharded = sharded_full = sharded16 = sharded16_full = sharded_pf = sharded_c4 = sharded_tr = sharded_a32 = None
    if args.total_crops > 0 and not args.no_extras and CB == 1:
        sharded_a32 = sharded_area32(args.total_crops)
        wl = ("BASELINE configs[3]: %d crops of %dx%d rays sharded crop i -> rank i mod %d, chunks of %d through BatchRefiner (reference losses + "
              "solver, HIP-graph replay), one all_gather of the result rows")
        sharded = sharded_section("exact float32 decoder (parity path); candidate reuse: the exact-f32 kernels run on the band candidates alone while a "
                                  "proven Lipschitz bound keeps them valid (bit-identical to the full-grid evaluation, audited)", torch.float32, True, H,
                                  args.total_crops, wl)
        if world == 1:        # every grid row every iteration (the r01-r04 figure; an eighth of the crops: 1.7 ms per crop-iteration)
            sharded_full = sharded_section("exact float32 decoder, every grid row every iteration (the r04 figure)", torch.float32, False, H,
                                           max(64, args.total_crops // 8), wl)
        sharded16 = sharded_section("float16 decoder = the reference's shipped precision (config_refine.ini:19), f32 everything else; candidate reuse: "
                                    "the half decoder runs on the band candidates alone while a proven Lipschitz bound keeps them valid (bit-identical "
                                    "to the full-grid evaluation, audited); two chunks in flight per rank", torch.float16, True, H, args.total_crops, wl, in_flight=2)
        if world == 1:
            sharded16_full = sharded_section("float16 decoder, every grid row every iteration (the r04 figure)", torch.float16, False, H, args.total_crops, wl)
        if world == 1:
            sharded_pf = sharded_section("float32_prefilter + candidate reuse: f16 pass proposes, exact f32 on everything consumed, run-time guard",
                                         "float32_prefilter", True, H, args.total_crops, wl)
            # ... and the same flow with the sphere tracer as the loop's renderer (float16 decoder; an eighth of the crops: the traced iteration
            # costs 4x the float16 splat path's)
            sharded_tr = sharded_section("sphere tracer as the loop's renderer, float16 decoder", torch.float16, False, H, max(world, args.total_crops // 8),
                                         wl, render="trace")
        if H == 256 and args.configs4_crops > 0 and world == 1:
            sharded_c4 = sharded_section("BASELINE configs[4] shape: 512x512 rays, float16 decoder on the f16 matrix cores, candidate reuse", torch.float16, True, 512,
                                         args.configs4_crops, "BASELINE configs[4]: %d crops of %dx%d rays, float16 DeepSDF decoder, sharded crop i -> rank "
                                         "i mod %d, chunks of %d through BatchRefiner, one all_gather", in_flight=2)

    # ---- labelled second line: pose-only refinement (BASELINE configs[1] says "pose-only refinement"; SURVEY.md 8d: "latent frozen -- MLP
    # result may be cached; state whether it was").  The HEADLINE above caches nothing.  Here the latent is fixed, so decoder, band and Jacobian
    # are evaluated once (BatchRenderer.freeze_shape) and a step is: pose -> re-projection -> splat -> backward to yaw and trans.
    def pose_only_setup():
        b2 = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), CB, device=dev)
        b2.freeze_shape = True
        b2.set_params(br.yaw, br.trans, br.latent)
        for _ in range(args.warmup + 1):
            b2.forward()
            b2.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)
        # the step as the refinement loop runs it: one HIP-graph replay (BatchRenderer.capture; five launches of 4-20 us each are
        # launch-bound from Python otherwise)
        b2.replay_step = b2.capture(lambda o: dict(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx))
        return b2

    def pose_only_run(b2):
        for _ in range(args.steps):
            b2.yaw.add_(1e-3)                                  # the pose moves every step, as under a solver; the latent does not
            b2.replay_step()

    pose_only = None
    if extras:
        res, err = timed_section(pose_only_setup, pose_only_run)
        if res is None:
            pose_only = {"error": err}
        else:
            b2, dt_p = res
            pose_only = {"workload": "pose-only crop-iteration: latent frozen, decoder/band/Jacobian evaluated ONCE and cached; per step pose -> "
                                     "re-projection -> splat -> backward to yaw/trans (%d crop(s) of %dx%d per GPU), one HIP-graph replay per step" % (CB, H, W),
                         "value": H * W * CB * world * args.steps / dt_p, "unit": "rays/s", "ms_per_step": dt_p / args.steps * 1e3,
                         "decoder_cached": True, "surfels": int(b2.cnt[0])}
            del b2
        res = None

    # ---- the splat pair at 64 crops per launch (BASELINE configs[2] shape), for roofline_splat (rank 0 only, a few steps)
    splat64 = None
    if rank == 0 and CB == 1 and H == 256 and extras:
        try:
            b64 = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 64, device=dev)
            p64 = {k: torch.from_numpy(v).to(dev) for k, v in crop_params(list(range(64))).items()}
            b64.set_params(p64["yaw"], p64["trans"], p64["latent"])
            o3, o1, ox = torch.ones(64, 3, H, W, device=dev), torch.ones(64, 1, H, W, device=dev), torch.ones(64, b64.cap, 3, device=dev)
            e64 = [{"jacobian": ev_pair(), "splat_fwd": ev_pair(), "splat_bwd": ev_pair()} for _ in range(4)]
            for e in e64:
                b64.forward(events=e)
                b64.backward(g_color=o3, g_mask=o1, g_normals=o3, g_xyzf=ox, events=e)
            torch.cuda.synchronize()
            ms = {k: float(np.mean([e[k][0].elapsed_time(e[k][1]) for e in e64[1:]])) for k in e64[0]}
            nb = 64.0 * H * W * 64 + 72.0 * float(b64.cnt.sum()) + 48.0 * float(b64.fcnt.sum())
            splat64 = {"crops_per_launch": 64, "fwd_ms": ms["splat_fwd"], "bwd_ms": ms["splat_bwd"], "algorithmic_bytes": nb,
                       "achieved": nb / ((ms["splat_fwd"] + ms["splat_bwd"]) * 1e-3) / 1e9, "jacobian_ms": ms["jacobian"],
                       "jacobian_tflops": 2.0 * macs * float(b64.cnt.sum()) / (ms["jacobian"] * 1e-3) / 1e12}
            splat64["frac"] = splat64["achieved"] / 8000.0
            del b64, o3, o1, ox
        except Exception as e:                                 # informational only
            splat64 = {"error": repr(e)[:200]}

    # the same crop-iteration with the alternative decoder arithmetics.  Informational -- the headline and the 1e-4 parity claim
    # are the exact-f32 path's.
    #   float16        half operands on the matrix cores, f32 accumulate (reference default precision, configs/config_refine.ini:19;
    #                  BASELINE configs[4]); everything else float32
    #   float32_split  every f32 operand as a hi/lo pair of halves, three f16 MFMAs per product: float32-equivalent results (passes the
    #                  float32 goldens at the float32 tolerances, tests/test_gpu_parity.py::test_split_decoder_*)
    def alt_decoder(precision, dtype_label, reuse=False, audit=True):
        def setup():
            d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
            d2.prefilter_reuse = reuse
            d2.prefilter_audit = audit
            d2 = d2.to(dev)
            b2 = sdflabel_amd.BatchRenderer(d2, D, K_for(H, W), (W, H), CB, device=dev)
            b2.set_params(br.yaw, br.trans, br.latent)
            ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for _ in range(args.warmup):
                b2.forward()
                b2.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)
            return b2, ev2

        def run(st):
            b2, ev2 = st
            for i in range(args.steps):
                b2.forward(mlp_events=ev2[i])
                b2.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)

        res, err = timed_section(setup, run)
        if res is None:
            return {"error": err}
        (b2, ev2), dt2 = res
        m2 = float(np.mean([a.elapsed_time(b) for a, b in ev2]))
        out = {"value": H * W * CB * world * args.steps / dt2, "unit": "rays/s", "ms_per_step": dt2 / args.steps * 1e3,
               "dtype": dtype_label, "decoder_forward_ms": m2}
        if getattr(b2, "prefilter", False):
            # the two-stage mode does not execute the 2*M*G flops of a full-grid pass in f32: no flop rate is quoted for it
            out.update({"decoder_forward_ms_covers": "f16 grid pass + candidate selection + exact-f32 sdf and Jacobian of the candidates",
                        "candidates": int(b2.ccnt[0]), "prefilter_margin": b2.margin, "f16_pass_max_deviation_at_calibration": b2.f16_error,
                        "guard": b2.prefilter_report(), "candidate_reuse": bool(b2.reuse),
                        "audited": bool(b2.audit), "audit_note": "every step a rotating 1/16 slice of the NON-candidate rows is evaluated a second time, with "
                        "float32-grade values (the error-compensated split kernel: within 2.4e-7 of the exact-f32 one; decoder.prefilter_audit_arith = 'float32' "
                        "takes the exact kernel); a band row the half pass never proposed counts a hard violation (r04)" if b2.audit else "audit off (the r03 behaviour): "
                        "the half pass is checked at the candidates only",
                        "lipschitz_latent_calibrated": getattr(b2, "lipschitz", None)})
        else:
            out.update({"decoder_forward_tflops": 2.0 * macs * G * CB / (m2 * 1e-3) / 1e12, "f16_mfma_peak_tflops": 2500.0})
        out.update({
                "surfels": int(b2.cnt[0]), "mask_pixels_differing_from_f32": float((b2.mask != br.mask).float().mean()),
                "max_abs_sdf_diff_vs_f32": float((b2.sdf - br.sdf).abs().max()),          # whole grid (prefilter: rows outside the candidates keep f16 values)
                "max_abs_sdf_diff_vs_f32_at_band_rows": float((b2.sdf[b2.idx[0, :int(b2.cnt[0])].long()] - br.sdf[b2.idx[0, :int(b2.cnt[0])].long()]).abs().max())
                                                        if int(b2.cnt[0]) > 0 else 0.0,
                "max_abs_color_diff_vs_f32": float((b2.color - br.color).abs().max())})
        return out

    f16 = split = prefilter = None
    if extras:
        f16 = alt_decoder(torch.float16, "f16 decoder / f32 rest")
        split = alt_decoder("float32_split", "f32 results from error-compensated f16 operand pairs (3 f16 MFMAs per product) / f32 rest")
    #   float32_prefilter  a float16 pass over the grid proposes candidates |sdf| < 0.03 + margin; band membership, sdf and Jacobian of the
    #                  band come from the exact-f32 kernels run on the candidates only (decoder_forward_ms spans both passes incl. the Jacobian)
    prefilter_reuse = None
    if extras:
        prefilter = alt_decoder("float32_prefilter", "exact f32 on the band candidates chosen by an f16 pass over the grid / f32 rest")
        #   ... and with the candidate set reused while the latent has moved less than margin / (4 lip) since the last f16 pass (here the latent
        #   does not move at all between the bench's steps, as under the 3e-5 learning rate of the refinement: the pass runs every 17th step)
        prefilter_reuse = alt_decoder("float32_prefilter", "as prefilter_decoder, f16 pass skipped while the candidate set is provably still valid",
                                      reuse=True)
        if prefilter is not None and "error" not in prefilter:
            un = alt_decoder("float32_prefilter", "unaudited", audit=False)
            prefilter["ms_per_step_without_audit"] = un.get("ms_per_step")

    # ---- sphere-tracing render mode (BASELINE.json's literal wording; NOT the reference's algorithm, no parity claim against it -- DESIGN.md 3.6;
    # its oracle is oracle/sdf_oracle.py::sphere_trace): one crop, `march steps` decoder evaluations per active ray with ballot compaction and the
    # looping tail kernel, forward + backward to yaw/trans/latent, all HIP kernels, no host synchronisation inside a render.
    # Default schedule: 4 samples per ray and pass from pass 10 on (speculative passes: accepted while inside the previous sample's safe sphere),
    # 16 from pass 14 on (the survivors re-packed 4 to a tile); hit pass (value + Jacobian at the hits) in the decoder's precision.  `_plain` = the
    # f16 march without cones and speculative passes, `_exact_polish` = f16 march with the hit pass in exact float32, `_nocone` = without the cone
    # phase.  Cone marching on 4x4-pixel tiles ahead of the per-ray march is the default since r04 (culled tiles' rays cost no evaluation;
    # ray_evaluations counts cones + rays).  `configs4_*` = BASELINE configs[4]'s shape (512x512 rays, 256-step budget, fp16 MLP on MFMA).
    # roofline_march: decoder evaluations of the march (counted on the device, speculative samples included) x 2 M FLOP / march time (events around sdfr_trace_march) against the MFMA
    # peak of the march's operand type; step_kernel_hbm: algorithmic bytes of the advance / compaction kernel per ray-step.
    sphere = None
    if rank == 0 and CB == 1 and extras:
        sphere = {}
        # (label, decoder precision, step budget, samples per pass (None: default schedule), hit pass, cone tile (None: default 4; 0: off), crop edge)
        for label, prec, steps, spec_k, polish, cone, size in (
                ("f16_64_steps", torch.float16, 64, None, None, None, H),                 # the default tracer
                ("f32_64_steps", torch.float32, 64, None, None, None, H),
                ("configs4_f16_512x512_256_steps", torch.float16, 256, None, None, None, 512),      # BASELINE configs[4]'s shape: 512x512 rays, 256-step budget, fp16 MLP
                ("f16_64_steps_nocone", torch.float16, 64, None, None, 0, H),              # without the cone phase (the r03 default)
                ("f32_64_steps_nocone", torch.float32, 64, None, None, 0, H),
                ("f16_64_steps_exact_polish", torch.float16, 64, None, "exact", None, H),
                ("f16_64_steps_plain", torch.float16, 64, 1, None, 0, H),                  # plain sphere tracing: no cones, no speculative passes
                ("f16_128_steps", torch.float16, 128, None, None, None, H)):                # (configs[1]'s step budget)
            try:
                Hs = Ws = size
                d3, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
                tr = sdflabel_amd.SphereTracer(d3.to(dev), K_for(Hs, Ws), (Ws, Hs), 1, steps=steps, device=dev, spec_k=spec_k, polish=polish,
                                                cone_block=cone)
                prm = [crop.yaw.detach().clone(), crop.trans.detach().clone().view(1, 3), crop.latent.detach().clone().view(1, -1)]
                o3, o1 = torch.ones(1, 3, Hs, Ws, device=dev), torch.ones(1, 1, Hs, Ws, device=dev)

                tr.yaw.copy_(prm[0].reshape(-1)); tr.trans.copy_(prm[1]); tr.latent.copy_(prm[2])      # inputs resident before the timed region

                def tstep(ev=None):
                    tr.render(events=ev)
                    tr.backward(g_color=o3, g_depth=o1, g_normals=o3)

                for _ in range(3):
                    tstep()
                torch.cuda.synchronize()
                nrep = 20
                t_ = time.perf_counter()
                for _ in range(nrep):
                    tstep()
                torch.cuda.synchronize()
                dt_t = (time.perf_counter() - t_) / nrep
                evs = [{"march": ev_pair()} for _ in range(5)]
                for e in evs:
                    tstep(e)
                torch.cuda.synchronize()
                march_ms = float(np.mean([e["march"][0].elapsed_time(e["march"][1]) for e in evs]))
                st3 = tr.stats()
                peak = 2500.0 if prec == torch.float16 else F32_MFMA_PEAK_TFLOPS
                tfl = 2.0 * macs * st3["ray_evaluations"] / (march_ms * 1e-3) / 1e12
                sphere[label] = {"value": Hs * Ws / dt_t, "unit": "rays/s", "rays": Hs * Ws, "ms_per_render_fwd_bwd": dt_t * 1e3, "march_steps": steps, "march_ms": march_ms,
                                 "hits": st3["hits"], "unresolved_after_last_step": st3["unresolved"], "ray_evaluations": st3["ray_evaluations"],
                                 "head_steps": tr.head_steps, "tail_rows": tr.tail_rows, "samples_per_ray_and_pass_in_the_looping_kernel": tr.spec_k,
                                 "speculative_from_pass": tr.spec_from if tr.spec_k > 1 else None,
                                 "second_level": {"samples": tr.spec_k2, "from_pass": tr.spec_from2} if tr.spec_k2 > tr.spec_k else None,
                                 "speculation_levels_from_pass_samples": [list(l) for l in tr.levels], "q_max": tr.q_max,
                                 "hit_pass": "float16 (decoder)" if tr.half_polish else "float32",
                                 "cone_marching": {"tile_px": tr.cone_block, "passes": tr.cone_steps, "cone_evaluations": st3.get("cone_evaluations"),
                                                   "culled_tiles": st3.get("culled_tiles")} if tr.cone_block else None,
                                 "roofline_march": {"bound": "mfma", "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak,
                                                    "flops": 2.0 * macs * st3["ray_evaluations"]},
                                 "max_abs_sdf_at_marched_hits": float(tr.hit_residual.abs().max())}
                # the advance / compaction kernel of the head steps against the HBM roofline: algorithmic bytes per active ray and step (read: pixel 4,
                # state 16, sdf 4, far 4; written for a surviving ray: pixel 4, state 16, decoder row 4 (L + 3)) and, from the PMC passes of an
                # earlier run of tools/sphere_pmc.sh (not measured in this run), the counter bytes and rate per launch
                from sdflabel_amd._lib import stored_traffic
                tj = stored_traffic(ROOT, "traffic_sphere_step.json")            # None unless measured on this tree's trace.hip
                if label == "f16_64_steps" and tj is not None:
                    sphere[label]["step_kernel_hbm"] = {"bound": "hbm", "algorithmic_bytes_per_ray_step": 28 + 20 + 4 * (tr.L + 3),
                                                        "achieved": tj.get("GBps"), "peak": 8000.0, "unit": "GB/s",
                                                        "frac": (tj.get("GBps") or 0.0) / 8000.0, "traffic": tj.get("hbm_bytes_per_launch"),
                                                        "duration_us": tj.get("duration_us"),
                                                        "note": "a 13 us launch: latency-bound, not bandwidth-bound -- the fraction is informational",
                                                        "traffic_source": "profiles/traffic_sphere_step.json (PMC passes of an earlier run; a 13 us launch: latency-, not bandwidth-bound)"}
                del tr, d3
            except Exception as e:
                sphere[label] = {"error": repr(e)[:200]}

    # the same crop-iteration through the drop-in boundary (rank 0 only, informational): sdflabel_amd.Decoder / Grid3D / Rasterer called exactly
    # as pipelines/optimizer.py:96-123,156 calls the reference's, the caller's own torch ops (normalize, cat, pose assembly, sums and their autograd)
    # included.  `launches`: GPU kernels / memsets / copies of ONE iteration, split by who issued them -- inside the library's entry points
    # (profiler ranges sdfr::*) or in the caller's code -- and the host synchronisations (device-to-host copies) of each side.
    def launch_census(fn):
        import tempfile
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        path = os.path.join(tempfile.mkdtemp(), "trace.json")
        prof.export_chrome_trace(path)
        ev = json.load(open(path))["traceEvents"]
        rng = sorted((e["ts"], e["ts"] + e.get("dur", 0)) for e in ev if e.get("cat") == "user_annotation" and str(e.get("name", "")).startswith("sdfr::"))
        inside = lambda ts: any(a <= ts <= b for a, b in rng)
        launch_ts = {}
        syncs = {"library": 0, "caller": 0}
        for e in ev:
            if e.get("cat") in ("cuda_runtime", "cuda_driver"):
                if "args" in e and e["args"].get("correlation") is not None:
                    launch_ts[e["args"]["correlation"]] = e["ts"]
                # .item() / .tolist(): a blocking device-to-host copy (hipMemcpyWithStream on ROCm) -- the host waits for the stream to drain
                if str(e.get("name", "")) in ("hipMemcpyWithStream", "cudaMemcpyAsync", "hipMemcpy", "hipStreamSynchronize", "cudaStreamSynchronize"):
                    syncs["library" if inside(e["ts"]) else "caller"] += 1
                # r04: the band count is read through a pinned asynchronous copy + an EVENT wait with the Jacobian already queued behind it: still
                # a point where the host blocks (counted here, separately), but the stream is not drained and the GPU keeps working
                if str(e.get("name", "")) in ("hipEventSynchronize", "cudaEventSynchronize"):
                    syncs["library_event_waits" if inside(e["ts"]) else "caller_event_waits"] = syncs.get("library_event_waits" if inside(e["ts"]) else "caller_event_waits", 0) + 1
        out = {"library_hip_kernels": 0, "library_torch_glue": 0, "caller_torch_ops": 0, "unattributed": 0}
        glue = {}
        for e in ev:
            if e.get("cat") in ("kernel", "gpu_memset", "gpu_memcpy"):
                ts = launch_ts.get(e.get("args", {}).get("correlation"))
                nm = str(e.get("name", ""))
                if ts is None:
                    out["unattributed"] += 1
                    continue
                lib = inside(ts)
                if lib:
                    hipk = "sdfr_" in nm
                    out["library_hip_kernels" if hipk else "library_torch_glue"] += 1
                    if not hipk:
                        short = nm.split("<")[0].split("(")[0][-60:]
                        glue[short] = glue.get(short, 0) + 1
                else:
                    out["caller_torch_ops"] += 1
        out["library_torch_glue_by_kernel"] = glue
        out["host_syncs"] = syncs
        out["profiler_ranges"] = len(rng)
        return out

    dropin = None
    if rank == 0 and extras:
        grid = sdflabel_amd.Grid3D(D, dev)
        renderer = sdflabel_amd.Rasterer(torch.from_numpy(K_for(H, W)), (W, H)).to(dev)
        for _ in range(30):                                  # (host-bound loop: allocator, Python and the clocks settle over the first iterations)
            crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nd = max(200, args.steps)                            # >= 0.5 s, like the headline's long_run
        for _ in range(nd):
            l2, _, _ = crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize()
        dt_d = (time.perf_counter() - t1) / nd
        dropin = {"value": H * W / dt_d, "unit": "rays/s", "ms_per_step": dt_d * 1e3, "steps": nd, "warmup": 30,
                  "device_memory_allocated_after_mb": torch.cuda.memory_allocated() / 1e6,
                  "loss_rel_diff_vs_batched": abs(float(l2) - float(loss)) / max(1.0, abs(float(loss)))}
        try:
            dropin["launches"] = launch_census(lambda: crop_iteration(dec, grid, renderer, crop))
        except Exception as e:                                 # informational only
            dropin["launches"] = {"error": repr(e)[:200]}

    if rank == 0:
        rays = H * W * CB * world * args.steps
        line = {
            "metric": "rendered rays/sec (fwd+bwd)", "value": rays / dt, "unit": "rays/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "rccl_version": rccl_version, "launcher": launcher, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: one" if (CB == 1 and H == 256) else ("BASELINE configs[4]-style: %d" % CB if H == 512 else "BASELINE configs[2]-style: %d" % CB)) +
                                   " %dx%d crop(s) per GPU, DeepSDF 8x512 on a 40^3 grid, fwd+bwd to yaw/trans/latent, decoder re-evaluated every step" % (H, W),
                       "crops_per_gpu": CB, "rays_per_crop": H * W, "grid_points": G, "surfels": int(n_surf),
                       "front_facing": int(n_front), "march_steps": None, "parallelism": "crop-parallel x%d" % world},
        }
        flops = 2.0 * macs * G * CB
        ach = flops / (mlp_ms * 1e-3) / 1e12
        traffic = None
        from sdflabel_amd._lib import stored_traffic
        tfile = stored_traffic(ROOT, "traffic_mlp_forward.json")        # None unless measured on THIS tree's kernel sources (hash inside)
        if tfile is not None and CB == 1:          # the committed PMC passes profiled the single-crop launch
            traffic = tfile.get("hbm_bytes_per_launch")
        line["roofline"] = {"kernel": "sdfr_mlp_kernel<float,32,2,2,8,2,1,2> (fused decoder forward on the grid, saves ReLU masks)", "bound": "mfma",
                            "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS,
                            "traffic": traffic, "traffic_source": "profiles/traffic_mlp_forward.json (rocprofv3 PMC passes FETCH_SIZE x2 + WRITE_SIZE of an "
                            "earlier run of this command, per launch; NOT measured in this run)" if traffic is not None else None,
                            "flops_per_launch": flops, "avg_launch_ms": mlp_ms,
                            "timing": "mean of %d event-bracketed launches in a separate pass after the timed region" % ev_steps}
        line["long_run"] = {"steps": long_steps, "seconds": dt_long, "value": H * W * CB * world * long_steps / dt_long, "unit": "rays/s",
                            "ms_per_step": dt_long / long_steps * 1e3, "note": "the same step timed over >= 200 steps and >= 0.5 s"}
        nbytes = 64.0 * H * W * CB + 72.0 * float(br.cnt.sum()) + 48.0 * float(br.fcnt.sum())          # 64 P + 72 N + 48 N_f per crop, fwd+bwd
        ach_s = nbytes / ((kms["splat_fwd"] + kms["splat_bwd"]) * 1e-3) / 1e9
        tsplat = stored_traffic(ROOT, "traffic_splat.json") or {}
        if splat64 is not None and "error" not in splat64:
            splat64["traffic"] = tsplat.get("crops_64")
        line["roofline_splat"] = {"kernel": "sdfr_splat_fwd_kernel<0> + sdfr_splat_bwd_kernel<0> (surfel splat / depth-softmax composite and its backward)",
                                  "bound": "hbm", "achieved": ach_s, "peak": 8000.0, "unit": "GB/s", "frac": ach_s / 8000.0, "traffic": tsplat.get("crops_1") if CB == 1 else None,
                                  "algorithmic_bytes_per_launch_pair": nbytes, "fwd_ms": kms["splat_fwd"], "bwd_ms": kms["splat_bwd"],
                                  "crops_per_launch": CB, "at_64_crops_per_launch": splat64,
                                  "traffic_source": "profiles/traffic_splat.json (PMC passes of an earlier run; not measured in this run)",
                                  "note": "north_star's '>= 40 % HBM roofline on the march kernel': NOT APPLICABLE to this kernel pair -- it is instruction-bound "
                                          "(candidate evaluation chains: ~250 cycles per candidate and sweep with the reference's exact division / square root; "
                                          "~3100 SIMD-cycles per surfel wave in the backward), moves 4.4 MB per crop and is 2 % of the step; counter traffic is at "
                                          "1.1x the algorithmic bytes at 64 crops per launch.  DESIGN.md 3.4"}
        line["jacobian"] = {"avg_launch_ms": kms["jacobian"], "tflops": 2.0 * macs * float(br.cnt.sum()) / (kms["jacobian"] * 1e-3) / 1e12,
                            "f32_mfma_peak_tflops": F32_MFMA_PEAK_TFLOPS, "rows": int(br.cnt.sum())}
        line["long_run_ms_per_step"] = dt_long / long_steps * 1e3          # (the same step over >= 200 steps / >= 0.5 s; ms_per_step above is over exactly --steps)
        line["dropin_api"] = dropin
        line["refine_demo"] = refine
        line["refine_demo_traced"] = refine_traced
        line["optimizer_mirror_varied_crops"] = varied
        line["refine_sharded"] = sharded
        line["refine_sharded_full_grid"] = sharded_full
        line["refine_sharded_float16"] = sharded16
        line["refine_sharded_float16_full_grid"] = sharded16_full
        line["refine_sharded_area32"] = sharded_a32
        line["refine_sharded_prefilter"] = sharded_pf
        line["refine_sharded_configs4"] = sharded_c4
        line["refine_sharded_traced"] = sharded_tr
        line["pose_only"] = pose_only
        line["sphere_trace"] = sphere
        line["world_size"] = world
        line["f16_decoder"] = f16
        line["split_decoder"] = split
        line["prefilter_decoder"] = prefilter
        line["prefilter_reuse_decoder"] = prefilter_reuse
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        line["per_rank_ms_per_step"] = per_rank_ms
        # everything measured goes to bench_extras.json (and one copy under gpurun_out/ when that scratch directory exists); stdout carries
        # ONE compact line with the contract keys only (a 24 KB line could not be parsed by the driver)
        extras_path = write_extras(line, args.extras)
        print(compact_line(line, extras_path), flush=True)
    if dist is not None:
        dist.barrier()                 # rank 0 has rank-0-only sections behind it (sphere tracing, drop-in census): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
