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
the launch stream inside the timed region, and `cpu_baseline`: the numpy oracle (oracle/sdf_oracle.py, a port of the
reference's dense algorithm) timed on the host cores for one crop-iteration of the same workload (rank 0, N=1 only).
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
    """The oracle timed on the host: one crop-iteration of the same workload, dense N x P formulation as the reference."""
    from oracle import sdf_oracle as O
    from tests._util import fitted_state
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    st, spec = fitted_state()
    layers = O.decoder_layers_from_state(st, spec)
    K = K_for(H, W)
    Kinv = np.linalg.inv(K).astype(np.float32)
    lat = np.array([0.3, -0.5, 0.8], np.float32)
    lat = lat / np.linalg.norm(lat)
    pts = O.generate_point_grid(D)
    t0 = time.perf_counter()
    inp = np.concatenate([np.broadcast_to(lat, (pts.shape[0], 3)), pts], 1).astype(np.float32)
    sdf, cache = O.decoder_forward(layers, spec, inp, want_cache=True)
    Jall = O.decoder_backward_inputs(layers, spec, inp, cache, np.ones_like(sdf))
    pm, _, nm, idx, n_hat = O.get_surface_points(pts, sdf, Jall[:, 3:], 0.03)
    pose = O.render_pose(0.6, [0.0, 0.0, 3.5])
    proj = O.project_in_2D(K, pose, pm, nm, nm, (W, H), output_nocs=True)
    v3, nc = proj["points_3d"].astype(np.float32), proj["normals_3d"].astype(np.float32)
    c_attr = ((proj["colors_3d"] + 1) / 2).astype(np.float32)
    t_mlp = time.perf_counter() - t0
    # dense splat + composite + backward on a quarter of the image (rows H*3/8 .. H*5/8, through the object), extrapolated x4:
    # the dense N x P formulation costs the same for every pixel
    sub = O.pixel_grid((W, H)).reshape(H, W, 2)[H * 3 // 8:H * 5 // 8].reshape(-1, 2)
    t1 = time.perf_counter()
    Wm = O.inside_surfel(Kinv, sub, v3, nc, diam=0.04)
    color = np.minimum((Wm.T @ c_attr).T, 1)
    mask = np.minimum(Wm.sum(0), 1)
    nimg = np.minimum((Wm.T @ ((nc + 1) / 2)).T, 1)
    del Wm
    P = sub.shape[0]
    g_v3, g_n, g_c = O.splat_backward(Kinv, (W, H), v3, nc, c_attr, np.ones((3, P), np.float32), np.ones((1, P), np.float32), None,
                                      np.ones((3, P), np.float32), grid_2d=sub)
    t_rast = (time.perf_counter() - t1) * (H * W / float(P))
    t2 = time.perf_counter()
    g_points, _, _, g_pose = O.project_backward_dcm(pose, pm, nm, g_v3, g_n, g_c * 0.5, output_nocs=True, filt_idx=proj["filt_idx"],
                                                    g_p3_filt=np.ones_like(proj["points_3d_filt"]))
    g_sdf, _ = O.get_surface_points_backward(sdf, n_hat, idx, g_points)
    g_lat = (Jall * g_sdf)[:, :3].sum(0)
    dt = t_mlp + t_rast + (time.perf_counter() - t2)
    assert color.shape[1] == P and mask.shape[0] == P and nimg.shape[1] == P
    assert np.isfinite(g_lat).all() and np.isfinite(g_pose).all()
    return {"value": H * W / dt, "unit": "rays/s", "cores": int(blas_threads), "kind": "port",
            "sample": "1 crop-iteration (fwd+bwd) of the bench workload (256x256 rays, D=40, N=%d surfels) with the numpy oracle, dense "
                      "N x P as the reference: decoder fwd + input-Jacobian on all 64000 grid points and projection timed in full "
                      "(%.1f s), dense splat/composite fwd+bwd timed on the central quarter of the image and extrapolated x4 (%.1f s); "
                      "BLAS matmuls on %d threads, elementwise passes single-threaded" % (pm.shape[0], t_mlp, t_rast, blas_threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--crops-per-gpu", type=int, default=1, help="crops refined together per rank (1 = BASELINE configs[1]; 64 = configs[2])")
    ap.add_argument("--crop-size", type=int, default=256, help="crop edge in pixels (256 = BASELINE configs[1..3]; 512 = configs[4], informational)")
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
    dist = None
    if world > 1 or "RANK" in os.environ:          # under torch.distributed.run: always take the distributed path
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod

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

    def step(ev=None):
        br.forward(mlp_events=ev)
        br.backward(g_color=ones3, g_mask=ones1, g_normals=ones3, g_xyzf=onesx)     # d/d(out) of the plain sums used as the loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()                  # no collector pause inside a timed region (the steps allocate nothing, but the interpreter may still run it)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
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

    res, err = timed_section(refine_setup, lambda st: st[0].optimize(iters))
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

    # the same crop-iteration with the alternative decoder arithmetics.  Informational -- the headline and the 1e-4 parity claim
    # are the exact-f32 path's.
    #   float16        half operands on the matrix cores, f32 accumulate (reference default precision, configs/config_refine.ini:19;
    #                  BASELINE configs[4]); everything else float32
    #   float32_split  every f32 operand as a hi/lo pair of halves, three f16 MFMAs per product: float32-equivalent results (passes the
    #                  float32 goldens at the float32 tolerances, tests/test_gpu_parity.py::test_split_decoder_*)
    def alt_decoder(precision, dtype_label):
        def setup():
            d2, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
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
                        "candidates": int(b2.ccnt[0]), "prefilter_margin": b2.margin, "f16_pass_max_deviation_at_calibration": b2.f16_error})
        else:
            out.update({"decoder_forward_tflops": 2.0 * macs * G * CB / (m2 * 1e-3) / 1e12, "f16_mfma_peak_tflops": 2500.0})
        out.update({
                "surfels": int(b2.cnt[0]), "mask_pixels_differing_from_f32": float((b2.mask != br.mask).float().mean()),
                "max_abs_sdf_diff_vs_f32": float((b2.sdf - br.sdf).abs().max()),          # whole grid (prefilter: rows outside the candidates keep f16 values)
                "max_abs_sdf_diff_vs_f32_at_band_rows": float((b2.sdf[b2.idx[0, :int(b2.cnt[0])].long()] - br.sdf[b2.idx[0, :int(b2.cnt[0])].long()]).abs().max())
                                                        if int(b2.cnt[0]) > 0 else 0.0,
                "max_abs_color_diff_vs_f32": float((b2.color - br.color).abs().max())})
        return out

    f16 = alt_decoder(torch.float16, "f16 decoder / f32 rest")
    split = alt_decoder("float32_split", "f32 results from error-compensated f16 operand pairs (3 f16 MFMAs per product) / f32 rest")
    #   float32_prefilter  a float16 pass over the grid proposes candidates |sdf| < 0.03 + margin; band membership, sdf and Jacobian of the
    #                  band come from the exact-f32 kernels run on the candidates only (decoder_forward_ms spans both passes incl. the Jacobian)
    prefilter = alt_decoder("float32_prefilter", "exact f32 on the band candidates chosen by an f16 pass over the grid / f32 rest")

    # the same crop-iteration through the drop-in boundary (rank 0 only, informational)
    dropin = None
    if rank == 0:
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
        line["dropin_api"] = dropin
        line["refine_demo"] = refine
        line["f16_decoder"] = f16
        line["split_decoder"] = split
        line["prefilter_decoder"] = prefilter
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
