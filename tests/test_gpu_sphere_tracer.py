"""GPU tests of the sphere-tracing render mode (BASELINE.json's literal wording).  NOT in the reference -- its renderer
splats surfels -- so there is no reference output; the mode's oracle is oracle/sdf_oracle.py::sphere_trace / sphere_trace_backward (numpy: the same
ray set-up, step rule, hit / exit tests, Newton polish and implicit-function gradient), checked here on > 2000 rays incl. grazing ones:
  * hit set equal wherever the oracle's closest decision was further than 1e-4 from its threshold (recorded per ray),
  * depth / NOCS colour / normals within 1e-4 on the non-grazing hits,
  * gradients of a random functional w.r.t. yaw, trans, latent within 1e-3 of the oracle's implicit-function restatement,
plus self-consistency, finite differences, the splat renderer as a cross-check (band thickness), the float16 march and batches."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from oracle import sdf_oracle as O
from tests._util import ASSET, K_for, fitted_state
from tests.test_gpu_parity import N, T

pytestmark = pytest.mark.gpu
DEV = "cuda"
YAW, TRANS, LAT = [0.6], [[0.05, -0.03, 3.5]], [[0.3, -0.5, 0.8]]


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


@pytest.fixture(scope="module")
def oracle_layers():
    st, spec = fitted_state()
    return O.decoder_layers_from_state(st, spec), spec


def _args(yaw=YAW, trans=TRANS, lat=LAT, grad=False):
    a = [torch.tensor(v, dtype=torch.float32, device=DEV) for v in (yaw, trans, lat)]
    return [t.requires_grad_(True) for t in a] if grad else a


@pytest.mark.parametrize("head_steps,tail_rows,spec_k,spec_k2", [(24, 4096, 1, 1), (0, 4096, 1, 1), (64, 0, 1, 1), (24, 4096, 4, 1), (0, 4096, 4, 1),
                                                                 (64, 0, 4, 1), (24, 4096, 4, 16), (0, 4096, 4, 16), (64, 0, 4, 8)])
def test_march_polish_images_and_gradients_against_the_oracle(dec, oracle_layers, head_steps, tail_rows, spec_k, spec_k2):
    """head 24 / tail 4096: the default (per-step launches while >= 4096 rays are active, then the looping tail kernel); head 0: EVERY ray is
    marched by the looping kernel alone; tail_rows 0: per-step launches only (until the speculative passes start, which exist in the looping
    kernel only).  spec_k = 4: from pass 16 on four samples per ray and pass; spec_k2 = 8 / 16: from pass 20 on the survivors are re-packed into
    tiles of 8 / 4 rays by a second launch of the looping kernel and take 8 / 16 samples per pass.  All schedules must reproduce the oracle."""
    layers, spec = oracle_layers
    H, W = 96, 128
    K = K_for(H, W)
    K[0, 2] += 9.0                                             # principal point off the image centre
    tr = sdflabel_amd.SphereTracer(dec, K, (W, H), 1, steps=64, device=DEV, head_steps=head_steps, tail_rows=tail_rows, spec_from=16, spec_k=spec_k,
                                  spec_from2=20, spec_k2=spec_k2)
    a = _args(grad=True)
    out = tr(*a)
    # ---- the oracle on a subset of > 2000 rays: every 2nd row and column (the object's silhouette crosses them: grazing rays included)
    ys, xs = np.meshgrid(np.arange(0, H, 2), np.arange(0, W, 2), indexing="ij")
    px = np.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    assert px.shape[0] > 2000
    lat = np.asarray(LAT[0], np.float32)
    latn = lat / np.sqrt((lat * lat).sum())
    pose = O.render_pose(YAW[0], TRANS[0])
    Kinv = np.linalg.inv(K).astype(np.float32)
    # (r04: cone marching on 4x4-pixel tiles is the tracer's default; the oracle takes the same cone phase)
    ref = O.sphere_trace(layers, spec, latn, pose, Kinv, px, steps=64,
                         spec_from=[(16, spec_k), (20, max(spec_k, spec_k2))] if spec_k > 1 else None, cone_block=4, cone_steps=tr.cone_steps,
                         cone_spec_k=tr.cone_spec_k, image_wh=(W, H))
    assert tr.cone_block == 4 and ref["cone_culled"].sum() > 200
    sel = (px[:, 1], px[:, 0])
    hit = N(out["mask"][0, 0])[sel] > 0
    safe = ref["margin"] > 1e-4
    assert ref["hit"].sum() > 500 and (ref["hit"] & ~ref["ok"]).sum() >= 1, "the sample must contain grazing hits"
    assert safe.mean() > 0.9                                   # (a cone decision within 1e-4 of its threshold marks all 16 rays of its tile)
    assert np.array_equal(hit[safe], ref["hit"][safe]), int((hit[safe] != ref["hit"][safe]).sum())
    good = safe & ref["hit"] & hit & ref["ok"]
    assert good.sum() > 250
    depth = N(out["depth"][0, 0])[sel]
    color = N(out["color"][0])[:, sel[0], sel[1]].T
    nrm = N(out["normals"][0])[:, sel[0], sel[1]].T
    assert np.abs(depth - ref["depth"])[good].max() < 1e-4
    assert np.abs(color - ref["color"])[good].max() < 1e-4
    # (a hit on a ReLU kink of the decoder may take its gradient from the other side of the kink in the two summation orders: isolated rays)
    dn = np.abs(nrm - ref["normals"])[good].max(1)
    assert np.median(dn) < 1e-6 and (dn > 1e-4).sum() <= 3, (np.median(dn), (dn > 1e-4).sum())
    st = tr.stats()
    assert st["hits"] == int((N(out["mask"]) > 0).sum()) and st["ray_evaluations"] > st["hits"]
    # ---- gradients of a random functional of the sampled rays' image values
    rng = np.random.default_rng(7)
    gC, gD, gN = rng.standard_normal((px.shape[0], 3)), rng.standard_normal(px.shape[0]), rng.standard_normal((px.shape[0], 3))
    use = (safe & ~(dn_all(nrm, ref) > 1e-4)).astype(np.float64)         # rays that both sides treat alike
    gC, gD, gN = gC * use[:, None], gD * use, gN * use[:, None]
    wC, wD, wN = torch.zeros(1, 3, H, W, device=DEV), torch.zeros(1, 1, H, W, device=DEV), torch.zeros(1, 3, H, W, device=DEV)
    ty, tx = torch.as_tensor(sel[0], device=DEV), torch.as_tensor(sel[1], device=DEV)
    wC[0][:, ty, tx] = T(gC.T.astype(np.float32))
    wD[0, 0][ty, tx] = T(gD.astype(np.float32))
    wN[0][:, ty, tx] = T(gN.T.astype(np.float32))
    ((out["color"] * wC).sum() + (out["depth"] * wD).sum() + (out["normals"] * wN).sum()).backward()
    g_pose, g_latn = O.sphere_trace_backward(ref, pose, gC, gD, gN)
    c, s = np.cos(YAW[0]), np.sin(YAW[0])
    dR = np.array([[-s, 0, c], [0, 0, 0], [-c, 0, -s]])
    g_yaw = float((g_pose[:3, :3] * dR).sum())
    nl = np.sqrt((lat * lat).sum())
    g_lat = (g_latn - latn * (latn.astype(np.float64) @ g_latn)) / nl
    for got, want in ((N(a[0].grad), [g_yaw]), (N(a[1].grad)[0], g_pose[:3, 3]), (N(a[2].grad)[0], g_lat)):
        want = np.asarray(want)
        assert np.abs(got - want).max() < 1e-3 * max(1.0, np.abs(want).max()), (got, want)


@pytest.mark.parametrize("sched,q_max", [([(12, 4), (15, 16)], 1.0), ([(6, 4), (8, 8), (10, 16), (12, 32), (13, 64)], 1.5), ([(4, 8), (7, 64)], 3.0)])
def test_march_against_the_oracle_on_the_second_decoder(sched, q_max):
    """the ellipsoid fit (curved surface, smooth normals; tools/fit_decoder.py --shape ellipsoid): speculative schedules -- the r03 two-level
    one, five levels up to 64 samples per ray and pass with the radius ratio clamped at 1.5 (r04: every level its own launch of the looping
    kernel's workgroup pool, survivors handed from list to list), two coarse levels with q_max 3 -- hit set / depth / colour / normals against
    the oracle marching the SAME schedule, on every 2nd pixel"""
    from sdflabel_amd.fixtures import ASSET_ELLIPSOID
    d, _ = sdflabel_amd.setup_dsdf(ASSET_ELLIPSOID + ".pt", precision=torch.float32)
    d = d.to(DEV)
    st, spec = fitted_state(ASSET_ELLIPSOID)
    layers = O.decoder_layers_from_state(st, spec)
    H, W = 96, 96
    K = K_for(H, W)
    tr = sdflabel_amd.SphereTracer(d, K, (W, H), 1, steps=64, device=DEV, spec_levels=sched, q_max=q_max)
    assert tr.levels == sched and tr.q_max == q_max
    out = tr.render(*_args())
    assert tr.stats()["unresolved"] == 0
    ys, xs = np.meshgrid(np.arange(0, H, 2), np.arange(0, W, 2), indexing="ij")
    px = np.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    lat = np.asarray(LAT[0], np.float32)
    latn = lat / np.sqrt((lat * lat).sum())
    ref = O.sphere_trace(layers, spec, latn, O.render_pose(YAW[0], TRANS[0]), np.linalg.inv(K).astype(np.float32), px, steps=64,
                         spec_from=sched, q_max=q_max, cone_block=4, cone_steps=tr.cone_steps, cone_spec_k=tr.cone_spec_k, image_wh=(W, H))
    sel = (px[:, 1], px[:, 0])
    hit = N(out["mask"][0, 0])[sel] > 0
    safe = ref["margin"] > 1e-4
    assert ref["hit"].sum() > 200 and safe.mean() > 0.9                # (more speculative decisions, more rays with one of them near its threshold)
    assert np.array_equal(hit[safe], ref["hit"][safe])
    good = safe & ref["hit"] & hit & ref["ok"]
    assert good.sum() > 150
    assert np.abs(N(out["depth"][0, 0])[sel] - ref["depth"])[good].max() < 1e-4
    assert np.abs(N(out["color"][0])[:, sel[0], sel[1]].T - ref["color"])[good].max() < 1e-4
    dn = np.abs(N(out["normals"][0])[:, sel[0], sel[1]].T - ref["normals"])[good].max(1)
    assert np.median(dn) < 1e-6 and (dn > 1e-4).sum() <= 3


def test_march_with_a_layernorm_decoder():
    """the reference's LayerNorm decoder variant (weight_norm=False; the ellipsoid fit): no looping kernel for it -- per-step launches of its own
    forward kernel, plain tracing -- same oracle, same tolerances"""
    from sdflabel_amd.fixtures import ASSET_ELLIPSOID_LN
    d, _ = sdflabel_amd.setup_dsdf(ASSET_ELLIPSOID_LN + ".pt", precision=torch.float32)
    d = d.to(DEV)
    st, spec = fitted_state(ASSET_ELLIPSOID_LN)
    layers = O.decoder_layers_from_state(st, spec)
    H, W = 64, 80
    K = K_for(H, W)
    tr = sdflabel_amd.SphereTracer(d, K, (W, H), 1, steps=64, device=DEV)
    assert tr.generic_march and tr.spec_k == 1
    a = _args(grad=True)
    out = tr(*a)
    ys, xs = np.meshgrid(np.arange(0, H, 2), np.arange(0, W, 2), indexing="ij")
    px = np.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    lat = np.asarray(LAT[0], np.float32)
    latn = lat / np.sqrt((lat * lat).sum())
    ref = O.sphere_trace(layers, spec, latn, O.render_pose(YAW[0], TRANS[0]), np.linalg.inv(K).astype(np.float32), px, steps=64,
                         cone_block=4, cone_steps=tr.cone_steps, cone_spec_k=tr.cone_spec_k, image_wh=(W, H))
    sel = (px[:, 1], px[:, 0])
    hit = N(out["mask"][0, 0])[sel] > 0
    safe = ref["margin"] > 1e-4
    assert ref["hit"].sum() > 100 and safe.mean() > 0.9
    assert np.array_equal(hit[safe], ref["hit"][safe])
    good = safe & ref["hit"] & hit & ref["ok"]
    assert good.sum() > 80
    assert np.abs(N(out["depth"][0, 0])[sel] - ref["depth"])[good].max() < 1e-4
    assert np.abs(N(out["color"][0])[:, sel[0], sel[1]].T - ref["color"])[good].max() < 1e-4
    # gradients flow (finite, non-zero) through the LayerNorm decoder's recomputing Jacobian
    (out["depth"].sum() + out["color"].sum()).backward()
    assert all(bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0 for t in a)


@pytest.mark.parametrize("block,size,cone_k,cone_steps", [(4, (94, 126), 1, 10), (8, (96, 128), 1, 10), (4, (94, 126), 4, 4), (8, (96, 128), 2, 6)])
def test_cone_marching_first_phase_against_the_oracle(dec, oracle_layers, block, size, cone_k, cone_steps):
    """cone_block: one ray per pixel tile first (tiles clipped by the image border at 94x126); culled tiles' rays are misses without an
    evaluation of their own, the others start where their cone stopped.  Against the oracle's cone_march + sphere_trace on every 2nd pixel:
    hit set on the safe rays, depth / colour 1e-4 on the non-grazing hits; against plain tracing: the same image, fewer evaluations.
    cone_k > 1 (r04): speculative cone passes, cone_k samples per cone and pass in fewer passes."""
    layers, spec = oracle_layers
    H, W = size
    K = K_for(H, W)
    K[0, 2] += 9.0
    kw = dict(steps=64, device=DEV, spec_from=16, spec_k=4, spec_from2=20, spec_k2=16)
    tr = sdflabel_amd.SphereTracer(dec, K, (W, H), 1, cone_block=block, cone_steps=cone_steps, cone_spec_k=cone_k, **kw)
    plain = sdflabel_amd.SphereTracer(dec, K, (W, H), 1, cone_block=0, **kw)
    a = _args(grad=True)
    out = tr(*a)
    ref_img = {k: v.clone() for k, v in plain.render(*_args()).items()}
    st_c, st_p = tr.stats(), plain.stats()
    assert st_c["culled_tiles"] > 0.3 * tr.cone.numel() and st_c["ray_evaluations"] < 0.75 * st_p["ray_evaluations"], (st_c, st_p)
    assert abs(st_c["hits"] - st_p["hits"]) <= 3 and st_c["unresolved"] <= 1
    flips = (out["mask"] != ref_img["mask"])
    assert int(flips.sum()) <= 3
    both = ((out["mask"] > 0) & (ref_img["mask"] > 0)).view(-1)
    dd = (out["depth"] - ref_img["depth"]).abs().view(-1)[both]
    assert float(dd.median()) < 1e-5 and float(torch.quantile(dd, 0.98)) < 1e-4 and float(dd.max()) < 5e-2
    ys, xs = np.meshgrid(np.arange(0, H, 2), np.arange(0, W, 2), indexing="ij")
    px = np.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    lat = np.asarray(LAT[0], np.float32)
    latn = lat / np.sqrt((lat * lat).sum())
    ref = O.sphere_trace(layers, spec, latn, O.render_pose(YAW[0], TRANS[0]), np.linalg.inv(K).astype(np.float32), px, steps=64,
                         spec_from=[(16, 4), (20, 16)], cone_block=block, cone_steps=cone_steps, cone_spec_k=cone_k, image_wh=(W, H))
    sel = (px[:, 1], px[:, 0])
    hit = N(out["mask"][0, 0])[sel] > 0
    safe = ref["margin"] > 1e-4
    assert ref["hit"].sum() > 400 and ref["cone_culled"].sum() > 500 and safe.mean() > 0.9
    assert np.array_equal(hit[safe], ref["hit"][safe]), int((hit[safe] != ref["hit"][safe]).sum())
    good = safe & ref["hit"] & hit & ref["ok"]
    assert good.sum() > 250
    assert np.abs(N(out["depth"][0, 0])[sel] - ref["depth"])[good].max() < 1e-4
    assert np.abs(N(out["color"][0])[:, sel[0], sel[1]].T - ref["color"])[good].max() < 1e-4
    (out["depth"].sum() + out["color"].sum()).backward()
    assert all(bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0 for t in a)


def dn_all(nrm, ref):
    return np.abs(nrm - ref["normals"]).max(1)


@pytest.mark.parametrize("spec_k", [1, 4])
def test_looping_tail_per_step_launches_and_tail_only_agree(dec, spec_k):
    """the march schedules apply the same step rule per ray -- with speculative passes too: which passes are speculative depends on the pass
    index alone; the decoder values differ in the last bits between the tile geometries (64-row tiles: 32x32x2 MFMA, 16-row tiles: 16x16x4,
    different k order), so: same hit set up to a handful of threshold rays, same depths to 2e-5"""
    H = W = 128
    outs = []
    for head_steps, tail_rows in ((24, 4096), (0, 4096), (64, 0), (5, 1 << 30)):
        tr = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=64, device=DEV, head_steps=head_steps, tail_rows=tail_rows, spec_k=spec_k)
        tr.render(*_args())
        outs.append((tr.hit_lam.clone(), tr.depth.clone(), tr.stats()))
    h0, d0, s0 = outs[0]
    for h, d, st in outs[1:]:
        differ = int(((h > 0) != (h0 > 0)).sum())
        both = (h > 0) & (h0 > 0)
        assert differ <= 5 and float(((d - d0).abs().view(-1) * both).max()) < 2e-5, (differ, float(((d - d0).abs().view(-1) * both).max()))
        assert abs(st["hits"] - s0["hits"]) <= 5 and abs(st["ray_evaluations"] - s0["ray_evaluations"]) <= 0.01 * s0["ray_evaluations"]
        assert abs(st["unresolved"] - s0["unresolved"]) <= 5


def test_speculative_passes_find_the_same_surface_and_resolve_the_creeping_rays(dec):
    """spec_k = 4 against plain sphere tracing: the same silhouette up to threshold rays, the same polished depths (the marched points differ
    within eps; one Newton step takes both to the same surface point to second order), hardly more decoder evaluations (the speculative samples
    of creeping rays are nearly all accepted), and no ray left unresolved at the step budget"""
    H = W = 128
    a = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=64, device=DEV, spec_k=1)
    b = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=64, device=DEV, spec_k=4)
    oa = {k: v.clone() for k, v in a.render(*_args()).items()}
    ob = b.render(*_args())
    sa, sb = a.stats(), b.stats()
    both = ((oa["mask"] > 0) & (ob["mask"] > 0)).view(-1)
    assert float((oa["mask"] != ob["mask"]).float().mean()) < 2e-3
    d = (oa["depth"] - ob["depth"]).abs().view(-1)[both]
    # (the few grazing hits are not polished: their marched points differ by up to eps / sin(incidence) along the ray)
    assert float(d.median()) < 1e-5 and float(torch.quantile(d, 0.98)) < 1e-4 and float(d.max()) < 5e-2
    assert sb["ray_evaluations"] < 1.25 * sa["ray_evaluations"], (sa, sb)     # (small crop: 4 samples per pass from pass 8, 16 from 11 on -- often wasted)
    assert sb["unresolved"] <= sa["unresolved"] and sb["unresolved"] <= 1, (sa, sb)
    # the speculative march needs about half the passes: with a budget of 36 it still resolves every ray, plain tracing does not
    a36 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=36, device=DEV, spec_k=1)
    b36 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=36, device=DEV, spec_k=4)
    a36.render(*_args()); b36.render(*_args())
    assert b36.stats()["unresolved"] <= 1 < a36.stats()["unresolved"], (a36.stats(), b36.stats())


def test_hits_lie_on_the_level_set_and_the_march_terminates(dec):
    H = W = 128
    st = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=96, device=DEV)
    out = st(*_args())
    stats = st.stats()
    assert stats["hits"] > 2000 and stats["unresolved"] <= 0.01 * H * W      # (grazing rays may still be creeping along the surface)
    assert float(st.hit_residual.abs().max()) < st.eps                       # the march stopped inside the tolerance
    # after the Newton polish the decoder vanishes at the hit points
    m = out["mask"][0, 0] > 0
    x = (out["color"][0].permute(1, 2, 0)[m] * 2 - 1) * torch.tensor([-1.0, 1.0, 1.0], device=DEV)      # NOCS colour -> object point
    latn = torch.nn.functional.normalize(torch.tensor(LAT[0], device=DEV), dim=0)
    sdf, _ = dec(torch.cat([latn.expand(x.shape[0], -1), x], 1).contiguous())
    r = N(sdf).reshape(-1)
    assert np.median(np.abs(r)) < 2e-5 and np.quantile(np.abs(r), 0.99) < 1e-3 and np.abs(r).max() < 5e-3, (np.median(np.abs(r)), np.quantile(np.abs(r), 0.99), np.abs(r).max())
    assert set(np.unique(N(out["mask"])).tolist()) == {0.0, 1.0}
    d = N(out["depth"][0, 0])[N(m)]
    assert d.min() > 2.0 and d.max() < 5.0
    nrm = N(out["normals"][0].permute(1, 2, 0)[m]) * 2 - 1
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-4 and (nrm[:, 2] < 0.2).mean() > 0.95     # camera-facing


def test_agrees_with_the_splat_renderer_up_to_the_band_thickness(dec):
    """cross-check only: the faithful path renders surfels of the |sdf| < 0.03 band of a 40^3 grid; the traced level set must give the same
    silhouette (up to the disc radius), the same depth (up to the band / disc size) and the same NOCS colours"""
    H = W = 128
    st = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=96, device=DEV)
    o = st(*_args())
    br = sdflabel_amd.BatchRenderer(dec, 40, K_for(H, W), (W, H), 1, device=DEV)
    s = br.forward(*_args())
    mt, ms = N(o["mask"][0, 0]) > 0, N(s["mask"][0, 0]) > 0
    iou = (mt & ms).sum() / (mt | ms).sum()
    assert iou > 0.85, iou
    both = mt & ms
    dd = np.abs(N(o["depth"][0, 0]) - N(s["depth"][0, 0]))[both]
    assert np.median(dd) < 0.02 and np.quantile(dd, 0.9) < 0.06, (np.median(dd), np.quantile(dd, 0.9))
    dc = np.abs(N(o["color"][0]) - N(s["color"][0])).max(0)[both]
    assert np.median(dc) < 0.02 and np.quantile(dc, 0.9) < 0.05, (np.median(dc), np.quantile(dc, 0.9))


@pytest.mark.parametrize("which,index,delta", [("trans", 2, 2e-3), ("trans", 0, 2e-3), ("yaw", 0, 2e-3), ("latent", 1, 2e-2)])
def test_gradients_match_finite_differences_on_the_common_hit_set(dec, which, index, delta):
    H = W = 96
    st = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=96, device=DEV)
    wts = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(DEV)

    def render(shift):
        a = _args(grad=(shift == 0))
        if shift != 0:
            k = {"yaw": 0, "trans": 1, "latent": 2}[which]
            a[k] = a[k].clone()
            a[k].view(-1)[index] += shift
        o = st(*a)
        return a, o

    a0, o0 = render(0.0)
    # the decoder is piecewise linear (ReLU kinks) and hits carry a residual of up to eps: a single central difference scatters by several
    # percent around the derivative (measured for yaw: 271 ... 317 around the analytic 300), so three step sizes are averaged
    pairs = [(render(+h)[1], render(-h)[1], h) for h in (2 * delta, delta, delta / 2)]
    common = (o0["mask"] > 0)
    for op, om, _ in pairs:
        common = common & (op["mask"] > 0) & (om["mask"] > 0)
    common = common.float()
    assert float(common.sum()) > 1500

    def functional(o):
        return (o["depth"] * common).sum() + (o["color"] * wts * common).sum()

    functional(o0).backward()                                   # (the tracer has rendered six more images since: the autograd path keeps its own state)
    g = {"yaw": a0[0].grad, "trans": a0[1].grad, "latent": a0[2].grad}[which].view(-1)[index]
    fds = [float((functional(op) - functional(om)) / (2 * h)) for op, om, h in pairs]
    fd = float(np.mean(fds))
    assert abs(float(g) - fd) < 0.08 * max(1.0, abs(fd)), (float(g), fds)


def test_half_operand_march_and_batches(dec):
    """float16 decoder on the march, hit pass exact or in half as well: same image up to half precision; a batch renders each crop as alone,
    gradients included (fixed-order sums per crop)"""
    H = W = 96
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    s32 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=64, device=DEV)
    a = s32(*_args())
    # polish "exact": the hit pass (value + Jacobian at the marched points) in float32 whatever the decoder; "decoder" (default): in half too
    for polish, tol_med, tol_n in (("exact", 2e-5, 2e-3), ("decoder", 1e-3, 2e-2)):
        s16 = sdflabel_amd.SphereTracer(d16.to(DEV), K_for(H, W), (W, H), 1, steps=64, device=DEV, polish=polish)
        a16 = _args(grad=True)
        b = s16(*a16)
        assert s16.half == 1 and s16.half_polish == (polish == "decoder") and float((a["mask"] != b["mask"]).float().mean()) < 0.01
        both = (a["mask"] > 0) & (b["mask"] > 0)
        dd = (a["depth"] - b["depth"]).abs()[both]
        # (the few grazing hits are not polished: their marched points differ by up to eps / sin(incidence) along the ray)
        assert float(torch.quantile(dd, 0.98)) < 5e-3 and float(dd.max()) < 5e-2 and float(dd.median()) < tol_med, \
            (polish, float(dd.max()), float(torch.quantile(dd, 0.98)), float(dd.median()))
        dn = (a["normals"] - b["normals"]).abs().amax(1, keepdim=True)[both]
        assert float(dn.median()) < tol_n, (polish, float(dn.median()))
        # gradients of a smooth functional: the half hit pass stays within half precision of the exact one
        a32 = _args(grad=True)
        o32 = s32(*a32)
        keep = both.float()
        ((o32["color"] * keep).sum() + (o32["depth"] * keep).sum()).backward()
        ((b["color"] * keep).sum() + (b["depth"] * keep).sum()).backward()
        for g32, g16 in zip(a32, a16):
            # (the two marches resolve a handful of silhouette rays differently: each carries ~1e-3 of this functional; 2.8e-2 measured)
            assert float((g32.grad - g16.grad).abs().max()) < 5e-2 * max(1.0, float(g32.grad.abs().max())), (polish, g32.grad, g16.grad)
    yaw, trans, lat = [0.6, -0.4], [[0.05, -0.03, 3.5], [0.1, 0.0, 3.0]], [[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]]
    s2 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 2, steps=64, device=DEV)
    wts = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(5)).to(DEV)
    o2 = s2(*_args(yaw, trans, lat))
    for i in range(2):
        a1 = _args(yaw[i:i + 1], trans[i:i + 1], lat[i:i + 1], grad=True)
        o1 = s32(*a1)
        # (the march switches from 64-row to 16-row decoder tiles when the TOTAL active count drops below tail_rows: in a batch a ray may see
        # the other tile geometry at a step -- decoder values equal to float rounding, not bit for bit)
        flips = (o1["mask"][0] != o2["mask"][i])
        keep = (~flips).float()
        assert int(flips.sum()) <= 3
        assert float(((o1["depth"][0] - o2["depth"][i]).abs() * keep).max()) < 1e-4 and float(((o1["color"][0] - o2["color"][i]).abs() * keep).max()) < 1e-4
        ((o1["color"] * wts[i:i + 1] * keep).sum() + (o1["depth"] * keep).sum()).backward()
        a2i = _args(yaw, trans, lat, grad=True)
        o2i = s2(*a2i)
        ((o2i["color"][i] * wts[i] * keep).sum() + (o2i["depth"][i] * keep).sum()).backward()
        for g1, g2 in zip(a1, a2i):
            ref = g1.grad[0]
            assert float((ref - g2.grad[i]).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), (g1.grad, g2.grad)
            other = g2.grad[1 - i]
            assert float(other.abs().max()) == 0.0                      # a crop's functional has no gradient in the other crop's parameters


def test_half_hit_pass_tile_geometries_agree():
    """the mask-fed half Jacobian at the hits on 64-row tiles (SDFR_JAC_MANY_ROWS, what the tracer launches for its thousands of hits) against
    the band geometry's 16-row tiles on the same masks: the same in-gradients up to the summation order of half products"""
    from sdflabel_amd import _lib
    H = W = 128
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    tr = sdflabel_amd.SphereTracer(d16.to(DEV), K_for(H, W), (W, H), 1, steps=64, device=DEV)
    tr.render(*_args())
    n_hits = int(tr.counters[6])
    assert tr.half_polish and n_hits > 3000
    L_, P = _lib.lib(), _lib.ptr
    n = H * W
    J64, f64_ = tr.J[:n_hits].clone(), tr.f0[:n_hits].clone()
    J16, f16_ = torch.zeros_like(tr.J), torch.zeros_like(tr.f0)
    with _lib.guard(tr.dev):
        _lib.check(L_.sdfr_mlp_jacobian(tr.handle.h, P(tr.rows), n, 1, P(tr.idx), n, P(tr.counters[6:7]), P(J16), P(f16_), P(tr.sdf), P(tr.mask_ws), 2,
                                        _lib.stream_ptr()), "sdfr_mlp_jacobian")
    assert torch.equal(f16_[:n_hits], f64_)                                      # the value is the forward's
    scale = float(J16[:n_hits].abs().max())
    assert float((J16[:n_hits] - J64).abs().max()) < 4e-3 * scale and float((J16[:n_hits] - J64).abs().mean()) < 2e-4 * scale
    # ... and against the exact-f32 Jacobian at the same rows: half precision
    Jx, fx = torch.zeros_like(tr.J), torch.zeros_like(tr.f0)
    with _lib.guard(tr.dev):
        _lib.check(L_.sdfr_mlp_jacobian(tr.handle.h, P(tr.rows), n, 1, P(tr.idx), n, P(tr.counters[6:7]), P(Jx), P(fx), None, None, 0, _lib.stream_ptr()),
                   "sdfr_mlp_jacobian")
    # (a handful of hits sit next to a ReLU kink of some hidden unit: the half pre-activation has the other sign there and the Jacobian jumps)
    dj = (Jx[:n_hits] - J64).abs().amax(1)
    assert float(torch.quantile(dj, 0.99)) < 3e-2 * scale and float(dj.median()) < 1e-3 * scale and float(dj.mean()) < 3e-3 * scale and float(dj.max()) < 0.3 * scale, \
        (float(torch.quantile(dj, 0.99)), float(dj.mean()), float(dj.max()), scale)
    assert float((fx[:n_hits] - f64_).abs().max()) < 5e-3


def test_half_march_on_every_tile_geometry_renders_the_same_surface():
    """the float16 march through its other kernels -- plain tracing (spec_k = 1: the looping kernel on 16-row tiles), non-uniform tiles
    (uniform_tiles=False: 16-row tiles of 16x16x32 products for thin cone passes, count-driven hand-over) -- against the default schedule (64- and
    128-row tiles of 32x32x16 products): the same hit set up to a few silhouette rays and the same depths to half precision.  (Late r04: the tiles
    of up to 64 rows keep two operand tiles in LDS -- every geometry of the half forward goes through here.)"""
    H, W = 96, 128
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    d16 = d16.to(DEV)
    ref = sdflabel_amd.SphereTracer(d16, K_for(H, W), (W, H), 1, steps=64, device=DEV)
    a = {k: v.clone() for k, v in ref.render(*_args()).items()}
    assert ref.stats()["unresolved"] == 0 and int((a["mask"] > 0).sum()) > 2000
    for kw in (dict(spec_k=1), dict(uniform_tiles=False), dict(spec_k=1, cone_block=0), dict(spec_levels=[(3, 8), (6, 32)], q_max=2.0)):
        tr = sdflabel_amd.SphereTracer(d16, K_for(H, W), (W, H), 1, steps=64, device=DEV, **kw)
        b = tr.render(*_args())
        assert tr.stats()["unresolved"] <= 2, (kw, tr.stats())
        flips = a["mask"] != b["mask"]
        assert int(flips.sum()) <= 6, (kw, int(flips.sum()))
        both = (a["mask"] > 0) & (b["mask"] > 0)
        dd = (a["depth"] - b["depth"]).abs()[both]
        assert float(dd.median()) < 1e-4 and float(torch.quantile(dd, 0.98)) < 5e-3, (kw, float(dd.median()), float(torch.quantile(dd, 0.98)))
        dc = (a["color"] - b["color"]).abs().amax(1, keepdim=True)[both]
        assert float(dc.median()) < 1e-4, (kw, float(dc.median()))
