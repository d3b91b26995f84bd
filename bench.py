"""bench.py -- rendered rays/sec (fwd+bwd) of the differentiable SDF renderer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env)

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

Extra objects on the JSON line: `roofline` for the dominant kernel (the fused decoder forward, MFMA-bound) timed with events on
the launch stream inside the timed region; `roofline_splat` (the splat forward+backward pair against the HBM roofline, at one crop and
at 64 crops per launch); `cpu_baseline`: the reference's dense algorithm as a multi-threaded torch-CPU port (oracle/torch_cpu_port.py,
pinned to the reference's golden G7) timed on the host cores for one full crop-iteration of the same workload (rank 0, N=1 only);
`refine_sharded`: BASELINE configs[3] -- `--total-crops` (default 1024) crops sharded crop i -> rank i mod N, refined in chunks of 64 by
BatchRefiner with the reference's losses and solver, one all_gather of the result rows: the strong-scaling figure of north_star
(time at N ranks / time at 1 rank); labelled second lines `pose_only`, `f16_decoder`, `split_decoder`, `prefilter_decoder`.
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


def K_for(h, w):
    f = 45.0 * h / 32.0
    return np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]], np.float32)


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
        g = torch.Generator().manual_seed(1 + index)
        jit = torch.rand(7, generator=g)
        self.yaw = (torch.tensor([0.6]) + 0.1 + 0.1 * jit[0:1]).to(dev).requires_grad_(True)
        self.trans = (torch.tensor([0.0, 0.0, 3.5]) + torch.tensor([0.1, 0.05, -0.3]) * jit[1:4]).to(dev).requires_grad_(True)
        self.latent = (torch.tensor([0.3, -0.5, 0.8]) + 0.2 * (jit[4:7] - 0.5)).to(dev).requires_grad_(True)


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
    from tests._util import fitted_state
    st, spec = fitted_state()
    decoder = TP.DecoderPort(st, spec)
    gp = TP.generate_point_grid(D).requires_grad_(True)
    K = torch.from_numpy(K_for(H, W))
    ncpu = os.cpu_count() or 1
    prev = torch.get_num_threads()
    out, n_surf = {}, 0

    def run():
        yaw = torch.tensor([0.6], requires_grad=True)
        trans = torch.tensor([0.0, 0.0, 3.5], requires_grad=True)
        lat = torch.tensor([0.3, -0.5, 0.8], requires_grad=True)
        t0 = time.perf_counter()
        rend, pts, n, loss = TP.crop_iteration(decoder, gp, K, (W, H), yaw, trans, lat)
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(yaw.grad).all() and torch.isfinite(trans.grad).all() and torch.isfinite(lat.grad).all())
        return dt, n

    # 8 threads = the survey's probe configuration; 32 = a quarter of a socket.  (Every hardware thread of the 256-thread GPU box measured
    # 1.08 k rays/s, 60 s per iteration: the dense passes are memory-bound and oversubscribe -- not timed by default)
    for threads in sorted({min(8, ncpu), min(32, ncpu)}, reverse=True):
        torch.set_num_threads(threads)
        if not out:
            run()                                   # one warm-up iteration (allocator, thread pool) at the first setting
        dt, n_surf = run()
        out[threads] = (H * W / dt, dt)
    torch.set_num_threads(prev)
    best = max(out, key=lambda k: out[k][0])
    return {"value": out[best][0], "unit": "rays/s", "cores": int(best), "kind": "port",
            "by_threads": {str(k): {"rays_per_s": v[0], "seconds_per_crop_iteration": v[1]} for k, v in out.items()},
            "host_cores": ncpu,
            "sample": "1 full crop-iteration (fwd+bwd to yaw/trans/latent) of the bench workload, all %dx%d rays, D=%d, N=%d surfels, dense "
                      "N x P formulation as the reference, torch CPU ops + autograd (oracle/torch_cpu_port.py); 1 warm-up + 1 timed "
                      "iteration per thread setting; value = the faster setting" % (H, W, D, n_surf)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--crops-per-gpu", type=int, default=1, help="crops refined together per rank (1 = BASELINE configs[1]; 64 = configs[2])")
    ap.add_argument("--crop-size", type=int, default=256, help="crop edge in pixels (256 = BASELINE configs[1..3]; 512 = configs[4], informational)")
    ap.add_argument("--total-crops", type=int, default=1024, help="crops of the sharded refinement (BASELINE configs[3]); 0 skips the section")
    ap.add_argument("--sharded-iters", type=int, default=10, help="refinement iterations per crop in the sharded section (the reference runs 60, "
                    "configs/config_refine.ini:15: pass 60 for the literal refine run; every iteration costs the same)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline loop (+ cpu_baseline): skip the informational sections")
    args = ap.parse_args()
    global H, W
    H = W = int(args.crop_size)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:                                              # each rank its share of the host cores
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    dist = None
    if world > 1 or "RANK" in os.environ:          # under torch.distributed.run: always take the distributed path
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod
        if dist.get_world_size() != world or args.gpus != world:
            raise SystemExit("--gpus %d, WORLD_SIZE %d, process group of %d ranks: they must agree" % (args.gpus, world, dist.get_world_size()))

    import sdflabel_amd
    from tests._util import ASSET
    if not os.path.isfile(sdflabel_amd.LIB_PATH):             # fresh checkout on the GPU box: compile the HIP library once (rank 0)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    dec = dec.to(dev)
    CB = args.crops_per_gpu
    from sdflabel_amd.parallel import shard_crops
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

    for _ in range(args.warmup):
        step()
    def ev_pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    events = [ev_pair() for _ in range(args.steps)]
    kev = [{"jacobian": ev_pair(), "splat_fwd": ev_pair(), "splat_bwd": ev_pair()} for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()                  # no collector pause inside a timed region (the steps allocate nothing, but the interpreter may still run it)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i], kev[i])
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    assert not br.overflow()
    n_surf, n_front = int(br.cnt[0]), int(br.fcnt[0])
    loss = br.color[0].sum() + br.mask[0].sum() + br.nimg[0].sum() + br.xyzf[0].sum()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the path's only exchange: per-crop result rows gathered once, outside the per-iteration critical path
        from sdflabel_amd.parallel import gather_crop_results
        res = torch.cat([br.color.sum(dim=(1, 2, 3)).view(CB, 1), br.g_yaw.view(CB, 1), br.g_trans, br.g_latent], dim=1).float()
        table = gather_crop_results(res, CB * world, rank, world)
        assert table.shape == (CB * world, 8) and bool(torch.isfinite(table).all())
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
        d_ = time.perf_counter() - t_
        gc.enable()
        if dist is not None:
            tt = torch.tensor([d_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d_ = float(tt.item())
        if not all_ok(err is None):
            return None, err or "failed on another rank"
        return (state, d_), None

    # the full refinement loop (reference: 60 iterations per crop, configs/config_refine.ini:15) with the reference's 2-D and 3-D losses and
    # its Adam/SGD step, device resident (sdflabel_amd.BatchRefiner); targets are rendered from the ground-truth pose (SURVEY.md 8 a-harness)
    iters = 60

    def refine_setup():
        rf = sdflabel_amd.BatchRefiner(dec, D, K_for(H, W), (H, W), CB, lidar_cap=4096, device=dev)
        gt = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=dev)
        o = gt.forward(torch.tensor([0.6], device=dev), torch.tensor([[0.0, 0.0, 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
        nfg = int(o["nf"][0])
        lidar = (o["xyzf"][0, :nfg] * 2.0)[::2].cpu().numpy()
        nocs_t = o["color"].expand(CB, 3, H, W).clone()
        p0 = {"yaw": torch.cat([c.yaw.detach() for c in crops]), "trans": torch.stack([c.trans.detach() for c in crops]),
              "scale": torch.full((CB,), 2.0), "latent": torch.stack([c.latent.detach() for c in crops])}
        rf.set_crops(p0, nocs_t, [lidar] * CB)
        rf.capture()
        rf.optimize(3)                                       # warm-up
        rf.set_crops(p0, nocs_t, [lidar] * CB)               # restart from the initial parameters
        rf.capture()
        return rf, p0["yaw"].to(dev).clone()

    res, err = timed_section(refine_setup, lambda st: st[0].optimize(iters)) if not args.no_extras else (None, "skipped (--no-extras)")
    if res is None:
        refine = {"error": err}
    else:
        (rf, y0), dt_r = res
        refine = {"value": CB * world / dt_r, "unit": "crops/s", "iterations_per_crop": iters, "ms_per_iteration": dt_r / iters * 1e3,
                  "crops": CB * world, "losses": "reference 2-D NOCS window loss + 3-D nearest-neighbour loss, Adam/SGD step, on device",
                  "yaw_error_before_after": [float((y0 - 0.6).abs().mean()), float((rf.yaw - 0.6).abs().mean())],
                  "crops_stepped_last_iteration": int(rf.stepped.sum())}
        del rf
    res = None

    # ---- BASELINE configs[3]: `--total-crops` crops sharded over the ranks (crop i -> rank i mod N), refined in chunks of 64 by BatchRefiner,
    # ONE all_gather of the per-crop result rows at the end.  Strong scaling: the total is fixed, so seconds(N=1) / seconds(N)
    # is the north_star's "x at 8 GPUs over 1 GPU on a 1024-crop batch".
    CHUNK = 64

    def crop_params(indices):
        ys, ts, ls = [], [], []
        for i in indices:                                      # the same per-crop jitter as Crop(i), generated on the host
            jit = torch.rand(7, generator=torch.Generator().manual_seed(1 + i))
            ys.append(0.6 + 0.1 + 0.1 * jit[0:1])
            ts.append(torch.tensor([0.0, 0.0, 3.5]) + torch.tensor([0.1, 0.05, -0.3]) * jit[1:4])
            ls.append(torch.tensor([0.3, -0.5, 0.8]) + 0.2 * (jit[4:7] - 0.5))
        return {"yaw": torch.cat(ys), "trans": torch.stack(ts), "scale": torch.full((len(indices),), 2.0), "latent": torch.stack(ls)}

    def sharded_setup():
        mine = shard_crops(args.total_crops, rank, world)
        rf = sdflabel_amd.BatchRefiner(dec, D, K_for(H, W), (H, W), CHUNK, lidar_cap=4096, device=dev)
        gt = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=dev)
        o = gt.forward(torch.tensor([0.6], device=dev), torch.tensor([[0.0, 0.0, 3.5]], device=dev), torch.tensor([[0.3, -0.5, 0.8]], device=dev))
        lidar = (o["xyzf"][0, :int(o["nf"][0])] * 2.0)[::2].cpu().numpy()
        nocs_t = o["color"].expand(CHUNK, 3, H, W).clone()
        params = crop_params(mine) if mine else None
        warm = crop_params(list(range(CHUNK)))
        rf.set_crops(warm, nocs_t, [lidar] * CHUNK)
        rf.capture()
        rf.optimize(2)                                         # warm-up (graph instantiation, allocator)
        return rf, mine, params, nocs_t, lidar

    def sharded_run(st):
        rf, mine, params, nocs_t, lidar = st
        rows, failure = [], None
        try:
            for c0 in range(0, len(mine), CHUNK):
                n = min(CHUNK, len(mine) - c0)
                sel = list(range(c0, c0 + n)) + [c0 + n - 1] * (CHUNK - n)      # a short last chunk is padded with copies of its last crop
                rf.set_crops({k: v[sel] for k, v in params.items()}, nocs_t, [lidar] * CHUNK)
                rf.optimize(args.sharded_iters)
                rows.append(rf.results()[0][:n])
            local = torch.cat(rows) if rows else torch.zeros((0, 5 + rf.L), device=dev)
        except Exception as e:                                 # a local failure still takes part in the collective (NaN rows), then reports
            failure = e
            local = torch.full((len(mine), 5 + rf.L), float("nan"), device=dev)
        st.append(gather_crop_results(local, args.total_crops, rank, world) if dist is not None else local)
        if failure is not None:
            raise failure

    sharded = None
    if args.total_crops > 0 and not args.no_extras and CB == 1:
        from sdflabel_amd.parallel import gather_crop_results
        res, err = timed_section(lambda: list(sharded_setup()), sharded_run)
        if res is None:
            sharded = {"error": err}
        else:
            st, dt_s = res
            table = st[-1]
            ok = tuple(table.shape) == (args.total_crops, 5 + st[0].L) and bool(torch.isfinite(table).all())
            sharded = {"workload": "BASELINE configs[3]: %d crops of %dx%d rays sharded crop i -> rank i mod %d, chunks of %d through BatchRefiner "
                                   "(reference losses + solver, HIP-graph replay), one all_gather of the result rows" % (args.total_crops, H, W, world, CHUNK),
                       "total_crops": args.total_crops, "iterations_per_crop": args.sharded_iters, "world_size": world, "seconds": dt_s,
                       "crop_iterations_per_s": args.total_crops * args.sharded_iters / dt_s,
                       "crops_per_s_at_this_iteration_count": args.total_crops / dt_s,
                       "mean_abs_yaw_error_after": float((table[:, 0] - 0.6).abs().mean()), "gathered_table_ok": ok,
                       "scaling": "strong (total crops fixed): speed-up at N ranks = seconds(N=1) / seconds(N)"}
            del st
        res = None

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
    if not args.no_extras:
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
    if rank == 0 and CB == 1 and H == 256 and not args.no_extras:
        try:
            b64 = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 64, device=dev)
            p64 = crop_params(list(range(64)))
            b64.set_params(p64["yaw"].to(dev), p64["trans"].to(dev), p64["latent"].to(dev))
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
    def alt_decoder(precision, dtype_label, reuse=False):
        def setup():
            d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
            d2.prefilter_reuse = reuse
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
    if not args.no_extras:
        f16 = alt_decoder(torch.float16, "f16 decoder / f32 rest")
        split = alt_decoder("float32_split", "f32 results from error-compensated f16 operand pairs (3 f16 MFMAs per product) / f32 rest")
    #   float32_prefilter  a float16 pass over the grid proposes candidates |sdf| < 0.03 + margin; band membership, sdf and Jacobian of the
    #                  band come from the exact-f32 kernels run on the candidates only (decoder_forward_ms spans both passes incl. the Jacobian)
    prefilter_reuse = None
    if not args.no_extras:
        prefilter = alt_decoder("float32_prefilter", "exact f32 on the band candidates chosen by an f16 pass over the grid / f32 rest")
        #   ... and with the candidate set reused while the latent has moved less than margin / (4 lip) since the last f16 pass (here the latent
        #   does not move at all between the bench's steps, as under the 3e-5 learning rate of the refinement: the pass runs every 17th step)
        prefilter_reuse = alt_decoder("float32_prefilter", "as prefilter_decoder, f16 pass skipped while the candidate set is provably still valid",
                                      reuse=True)

    # ---- sphere-tracing render mode (BASELINE.json's literal wording; NOT the reference's algorithm, no parity claim -- DESIGN.md 3.6):
    # one crop, `march steps` decoder evaluations per active ray with ballot compaction, forward + backward to yaw/trans/latent
    sphere = None
    if rank == 0 and CB == 1 and not args.no_extras:
        sphere = {}
        for label, prec, steps in (("f32_64_steps", torch.float32, 64), ("f16_64_steps", torch.float16, 64), ("f16_128_steps", torch.float16, 128)):
            try:
                d3, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=prec)
                tr = sdflabel_amd.SphereTracer(d3.to(dev), K_for(H, W), (W, H), 1, steps=steps, device=dev)
                prm = [crop.yaw.detach().clone().requires_grad_(True), crop.trans.detach().clone().view(1, 3).requires_grad_(True),
                       crop.latent.detach().clone().view(1, -1).requires_grad_(True)]

                def tstep():
                    for p_ in prm:
                        p_.grad = None
                    o_ = tr(*prm)
                    (o_["depth"].sum() + o_["color"].sum() + o_["normals"].sum()).backward()
                    return o_

                for _ in range(2):
                    tstep()
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                nrep = 5
                for _ in range(nrep):
                    o_ = tstep()
                torch.cuda.synchronize()
                dt_t = (time.perf_counter() - t_) / nrep
                sphere[label] = {"value": H * W / dt_t, "unit": "rays/s", "ms_per_render_fwd_bwd": dt_t * 1e3, "march_steps": steps, "march_steps_run": int(tr.steps_run),
                                 "rays_entering_the_object_cube": int(tr.n_entered), "hits": int(tr.n_hit),
                                 "unresolved_after_last_step": int(tr.n_unresolved), "max_abs_sdf_at_marched_hits": float(tr.hit_residual.abs().max())}
                del tr, d3
            except Exception as e:
                sphere[label] = {"error": repr(e)[:200]}

    # the same crop-iteration through the drop-in boundary (rank 0 only, informational)
    dropin = None
    if rank == 0 and not args.no_extras:
        grid = sdflabel_amd.Grid3D(D, dev)
        renderer = sdflabel_amd.Rasterer(torch.from_numpy(K_for(H, W)), (W, H)).to(dev)
        for _ in range(3):
            crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nd = max(5, args.steps // 3)
        for _ in range(nd):
            l2, _, _ = crop_iteration(dec, grid, renderer, crop)
        torch.cuda.synchronize()
        dt_d = (time.perf_counter() - t1) / nd
        dropin = {"value": H * W / dt_d, "unit": "rays/s", "ms_per_step": dt_d * 1e3,
                  "loss_rel_diff_vs_batched": abs(float(l2) - float(loss)) / max(1.0, abs(float(loss)))}

    if rank == 0:
        rays = H * W * CB * world * args.steps
        line = {
            "metric": "rendered rays/sec (fwd+bwd)", "value": rays / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: single" if (CB == 1 and H == 256) else ("BASELINE configs[4]-style: %d" % CB if H == 512 else "BASELINE configs[2]-style: %d" % CB)) + " %dx%d crop per GPU, DeepSDF 8x512 decoder on a 40^3 grid, " % (H, W) +
                                   "fwd+bwd to yaw/trans/latent (BatchRenderer, B=%d), decoder re-evaluated every step" % CB,
                       "crops_per_gpu": CB, "rays_per_crop": H * W, "grid_points": G, "surfels": int(n_surf),
                       "front_facing": int(n_front), "march_steps": None, "parallelism": "crop-parallel x%d" % world},
        }
        flops = 2.0 * macs * G * CB
        ach = flops / (mlp_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_mlp_forward.json")
        if os.path.isfile(tpath) and CB == 1:      # the committed PMC passes profiled the single-crop launch
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        line["roofline"] = {"kernel": "sdfr_mlp_kernel<float,32,2,2,8,2,1,2> (fused decoder forward on the grid, saves ReLU masks)", "bound": "mfma",
                            "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS,
                            "traffic": traffic, "flops_per_launch": flops, "avg_launch_ms": mlp_ms}
        nbytes = 64.0 * H * W * CB + 72.0 * float(br.cnt.sum()) + 48.0 * float(br.fcnt.sum())          # 64 P + 72 N + 48 N_f per crop, fwd+bwd
        ach_s = nbytes / ((kms["splat_fwd"] + kms["splat_bwd"]) * 1e-3) / 1e9
        tsp = os.path.join(ROOT, "profiles", "traffic_splat.json")
        tsplat = json.load(open(tsp)) if os.path.isfile(tsp) else {}
        if splat64 is not None and "error" not in splat64:
            splat64["traffic"] = tsplat.get("crops_64")
        line["roofline_splat"] = {"kernel": "sdfr_splat_fwd_kernel<0> + sdfr_splat_bwd_kernel<0> (surfel splat / depth-softmax composite and its backward)",
                                  "bound": "hbm", "achieved": ach_s, "peak": 8000.0, "unit": "GB/s", "frac": ach_s / 8000.0, "traffic": tsplat.get("crops_1") if CB == 1 else None,
                                  "algorithmic_bytes_per_launch_pair": nbytes, "fwd_ms": kms["splat_fwd"], "bwd_ms": kms["splat_bwd"],
                                  "crops_per_launch": CB, "at_64_crops_per_launch": splat64,
                                  "note": "latency-bound at one crop (candidate evaluation chains, not bytes); see DESIGN.md 3.4"}
        line["jacobian"] = {"avg_launch_ms": kms["jacobian"], "tflops": 2.0 * macs * float(br.cnt.sum()) / (kms["jacobian"] * 1e-3) / 1e12,
                            "f32_mfma_peak_tflops": F32_MFMA_PEAK_TFLOPS, "rows": int(br.cnt.sum())}
        line["dropin_api"] = dropin
        line["refine_demo"] = refine
        line["refine_sharded"] = sharded
        line["pose_only"] = pose_only
        line["sphere_trace"] = sphere
        line["world_size"] = world
        line["f16_decoder"] = f16
        line["split_decoder"] = split
        line["prefilter_decoder"] = prefilter
        line["prefilter_reuse_decoder"] = prefilter_reuse
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
