"""Device-memory hygiene of the product paths (r03): loops that the reference's pipeline runs thousands of times must release every
per-iteration buffer without the help of Python's cycle collector (a cycle through a tensor's C++ base pointer is invisible to it)."""
import gc

import pytest
import torch

import sdflabel_amd
from tests._util import ASSET, K_for, gold
from tests.test_gpu_parity import T

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _steady(fn, warm=3, n=12, limit=4 << 20):
    gc.collect()
    gc.disable()
    try:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown < limit, "device memory grew by %.1f MB over %d calls" % (grown / 1e6, n)


@pytest.mark.parametrize("precision", [torch.float32, torch.float16])
def test_sphere_tracer_autograd_calls(precision):
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
    H = W = 96
    tr = sdflabel_amd.SphereTracer(d.to(DEV), K_for(H, W), (W, H), 1, steps=48, device=DEV)

    def fn():
        a = [torch.tensor(v, dtype=torch.float32, device=DEV, requires_grad=True) for v in ([0.6], [[0.05, -0.03, 3.5]], [[0.3, -0.5, 0.8]])]
        o = tr(*a)
        (o["depth"].sum() + o["color"].sum()).backward()

    _steady(fn)


def test_standalone_primitives():
    from sdflabel_amd.renderer import primitives as PR
    z = gold("g13_primitives.npz")
    W, H = [int(v) for v in z["res"]]
    K = T(z["K"])
    r = sdflabel_amd.Rasterer(K, (W, H)).to(DEV)
    uv = T(z["uv"])

    def fn():
        p = T(z["points"]).requires_grad_(True)
        n = T(z["normals"]).requires_grad_(True)
        w = PR.inside_surfel(K, r.grid, uv, p, n, diam=0.04, softclamp=False, add_bg=False)
        w[:, 0].sum().backward()

    _steady(fn)


def test_batch_renderer_and_refiner_steps():
    from sdflabel_amd.fixtures import crop_params, synthetic_targets
    d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d32 = d32.to(DEV)
    D, H, W, B = 24, 64, 64, 3
    K = K_for(H, W)
    br = sdflabel_amd.BatchRenderer(d32, D, K, (H, W), B, device=DEV)
    prm = crop_params(list(range(B)))
    yaw, trans, lat = T(prm["yaw"]).view(B), T(prm["trans"]).view(B, 3), T(prm["latent"]).view(B, -1)
    g3, g1 = torch.ones(B, 3, H, W, device=DEV), torch.ones(B, 1, H, W, device=DEV)

    def step():
        br.forward(yaw, trans, lat)
        br.backward(g_color=g3, g_mask=g1)

    _steady(step)
    nocs1, lidar = synthetic_targets(d32, D, K, H, W, DEV)
    rf = sdflabel_amd.BatchRefiner(d32, D, K, (H, W), B, lidar_cap=4096, device=DEV)

    def refine():
        rf.set_crops(prm, nocs1.expand(B, 3, H, W), [lidar] * B)
        rf.optimize(3)
        rf.results()

    _steady(refine)


def test_optimizer_mirror_calls():
    """one Optimizer object per crop, as refine_css.py:203 constructs them: the refiner (and its captured graph) is shared, nothing piles up"""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    dsdf, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    dsdf = dsdf.to(DEV)
    grid = sdflabel_amd.Grid3D(D, DEV)
    K, nocs = T(z["K"]), T(z["nocs_target"])

    def fn():
        params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
        Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5}).optimize(3, nocs, z["lidar"], dsdf, grid, K, (H, W))

    _steady(fn, warm=2, n=6)
