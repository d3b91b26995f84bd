"""GPU tests of the tile-binned splat forward (count -> scan -> fill lists of surfels per 8x8 pixel tile): identical bits to the
all-boxes scan it replaces, and correct fall-backs when the lists do not fit."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from sdflabel_amd import _lib
from oracle import sdf_oracle as O
from tests._util import ASSET, K_for
from tests.test_gpu_parity import N, T, images_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
BINS = 512


def _splat(prim_flags, K, Kinv, p, n, attr, W, H, B=1, cnt=None):
    L = _lib.lib()
    cap = p.shape[-2]
    ws = _lib.splat_ws(B, cap, W, H, DEV)
    ws.fill_(-7)
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=DEV)
    color, mask, depth, nimg, aux = f(B, 3, H, W), f(B, 1, H, W), f(B, 1, H, W), f(B, 3, H, W), f(B, H * W, 4)
    _lib.check(L.sdfr_splat_forward(prim_flags, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p), _lib.ptr(n), _lib.ptr(attr), None, None, None, None, B,
                                    cap, _lib.ptr(cnt), W, H, 0.04, 150.0, _lib.ptr(ws), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth),
                                    _lib.ptr(nimg), _lib.ptr(aux), _lib.stream_ptr()), "sdfr_splat_forward")
    return color, mask, depth, nimg, aux, ws


def _surfels(rng, n, spread, z0, z1):
    p = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(z0, z1, n)], 1).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32) * 0.3 + np.array([0, 0, -1], np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return p, nrm, rng.uniform(0, 1, (n, 3)).astype(np.float32)


@pytest.mark.parametrize("H,W,n", [(64, 64, 900), (72, 100, 2500), (256, 256, 3000)])
def test_binned_forward_is_bitwise_the_unbinned_scan(H, W, n):
    rng = np.random.default_rng(H + n)
    p, nrm, col = _surfels(rng, n, 0.9, 3.0, 4.0)
    K = K_for(H, W)
    Kt, Ki = T(K).view(1, 9), T(np.linalg.inv(K).astype(np.float32)).view(1, 9)
    a = _splat(BINS, Kt, Ki, T(p)[None], T(nrm)[None], T(col)[None], W, H)
    b = _splat(0, Kt, Ki, T(p)[None], T(nrm)[None], T(col)[None], W, H)
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert float(a[1].sum()) > 50
    # the lists: valid flag set, total = sum over surfels of the tiles their boxes overlap, offsets ascending
    Tn = ((W + 7) // 8) * ((H + 7) // 8)
    ws = N(a[5])
    boxes = ws[:n * 4].reshape(n, 4)
    toff = ws[n * 4:n * 4 + Tn + 2]
    ok = boxes[:, 0] <= boxes[:, 2]
    want = int((((boxes[ok, 2] >> 3) - (boxes[ok, 0] >> 3) + 1) * ((boxes[ok, 3] >> 3) - (boxes[ok, 1] >> 3) + 1)).sum())
    assert toff[Tn + 1] == 1 and toff[Tn] == want and (np.diff(toff[:Tn + 1]) >= 0).all() and toff[0] == 0
    lst = ws[n * 4 + Tn + 2:n * 4 + Tn + 2 + want]
    assert lst.min() >= 0 and lst.max() < n


def test_bin_list_overflow_falls_back_to_the_scan_and_matches_the_oracle():
    """surfels so close to the camera that each covers most of the image: more list entries than 32 per surfel -> the crop is flagged
    unbinned and every tile scans all boxes (and here also overflows its 1024-entry LDS list? no: N is small) -- same images"""
    rng = np.random.default_rng(8)
    H, W, n = 96, 96, 40
    p, nrm, col = _surfels(rng, n, 0.01, 0.05, 0.06)          # disc radius 0.04 at z = 0.05: covers ~everything
    K = K_for(H, W)
    Kinv = np.linalg.inv(K).astype(np.float32)
    a = _splat(BINS, T(K).view(1, 9), T(Kinv).view(1, 9), T(p)[None], T(nrm)[None], T(col)[None], W, H)
    Tn = ((W + 7) // 8) * ((H + 7) // 8)
    toff = N(a[5])[n * 4:n * 4 + Tn + 2]
    assert toff[Tn + 1] == 0 and toff[Tn] > 32 * n                    # overflow recorded, lists not used
    b = _splat(0, T(K).view(1, 9), T(Kinv).view(1, 9), T(p)[None], T(nrm)[None], T(col)[None], W, H)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    Wm, aux = O.inside_surfel(Kinv, O.pixel_grid((W, H)), p, nrm, diam=0.04, want_aux=True)
    images_close(N(a[0][0]), np.minimum((Wm.T @ col).T, 1).reshape(3, H, W), aux)
    images_close(N(a[1][0]), np.minimum(Wm.sum(0), 1).reshape(1, H, W), aux)


def test_binned_ragged_batch_with_empty_and_full_crops():
    """device-side counts: crop 0 empty, crop 1 partly filled, crop 2 full -- each equals the same crop splatted alone"""
    rng = np.random.default_rng(3)
    H, W, cap = 64, 80, 700
    K = K_for(H, W)
    Kt = T(np.tile(K.reshape(1, 9), (3, 1))); Ki = T(np.tile(np.linalg.inv(K).astype(np.float32).reshape(1, 9), (3, 1)))
    p, nrm, col = _surfels(rng, 3 * cap, 0.8, 3.0, 4.0)
    p, nrm, col = T(p).view(3, cap, 3), T(nrm).view(3, cap, 3), T(col).view(3, cap, 3)
    cnt = torch.tensor([0, 311, cap], dtype=torch.int32, device=DEV)
    a = _splat(BINS, Kt, Ki, p, nrm, col, W, H, B=3, cnt=cnt)
    assert float(a[0][0].abs().max()) == 0.0 and float(a[1][0].abs().max()) == 0.0
    for b, c in ((1, 311), (2, cap)):
        one = _splat(BINS, Kt[b:b + 1], Ki[b:b + 1], p[b:b + 1, :c].contiguous(), nrm[b:b + 1, :c].contiguous(), col[b:b + 1, :c].contiguous(), W, H)
        for x, y in zip(a[:5], one[:5]):
            assert torch.equal(x[b], y[0]), b


def _splat_any(prim, K, Kinv, p, n, attr, W, H, B, uv=None, znorm=None, bg=None, bg_logit=None, diam=0.04):
    L = _lib.lib()
    cap = p.shape[-2]
    ws = _lib.splat_ws(B, cap, W, H, DEV)
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=DEV)
    outs = f(B, 3, H, W), f(B, 1, H, W), f(B, 1, H, W), f(B, 3, H, W), f(B, H * W, 4)
    no_dn = bg is not None           # the reference rejects depth / normals together with a background (rasterer.py:133,139)
    _lib.check(L.sdfr_splat_forward(prim, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p), _lib.ptr(n), _lib.ptr(attr), _lib.ptr(uv), _lib.ptr(znorm),
                                    _lib.ptr(bg), _lib.ptr(bg_logit), B, cap, None, W, H, diam, 150.0, _lib.ptr(ws), _lib.ptr(outs[0]),
                                    _lib.ptr(outs[1]), None if no_dn else _lib.ptr(outs[2]), None if no_dn else _lib.ptr(outs[3]),
                                    _lib.ptr(outs[4]), _lib.stream_ptr()), "sdfr_splat_forward")
    return outs[:2] + (outs[4],) if no_dn else outs


@pytest.mark.parametrize("case", ["sparse", "dense", "long_list", "list_overflow", "partial_tiles"])
@pytest.mark.parametrize("bins", [0, BINS])
def test_wave_per_tile_launch_is_bitwise_the_wave_per_share_launch(case, bins):
    """from 16384 tiles per launch the forward runs one wave per tile walking the 8 candidate shares in turn (many crops) instead of one wave
    per share: same share partition, same merge order -> the same bits.  sparse: <= 64 candidates per tile (kept resident); dense: hundreds
    (staged share by share, two rounds); long_list: 1300 candidates in a tile (r06: a list of 16-bit slots, the dense binned tile scans instead of
    rank-sorting); list_overflow: > 3072 candidates (every surfel walked, coverage re-evaluated); partial_tiles: image edges
    that are no multiples of 8."""
    H, W, n, spread, z0, z1 = {"sparse": (256, 256, 3000, 0.9, 3.0, 4.0), "dense": (64, 64, 700, 0.05, 0.3, 0.4),
                               "list_overflow": (64, 64, 3300, 0.01, 0.05, 0.06), "long_list": (64, 64, 1300, 0.01, 0.05, 0.06), "partial_tiles": (60, 100, 500, 0.8, 3.0, 4.0)}[case]
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    B = 16384 // tiles + 1
    rng = np.random.default_rng(21)
    p, nrm, col = _surfels(rng, n, spread, z0, z1)
    K = K_for(H, W)
    Kt, Ki = T(K).view(1, 9), T(np.linalg.inv(K).astype(np.float32)).view(1, 9)
    pt, nt, ct = T(p)[None], T(nrm)[None], T(col)[None]
    one = _splat_any(bins, Kt, Ki, pt, nt, ct, W, H, 1)
    rep = lambda t: t.expand(B, *t.shape[1:]).contiguous()
    many = _splat_any(bins, rep(Kt), rep(Ki), rep(pt), rep(nt), rep(ct), W, H, B)
    assert float(one[1].sum()) > 50
    for x, y in zip(many, one):
        for b in (0, B // 2, B - 1):
            assert torch.equal(x[b], y[0]), (case, b)


@pytest.mark.parametrize("prim", [1, 2])
@pytest.mark.parametrize("use_bg", [False, True])
def test_wave_per_tile_launch_secondary_primitives_and_background(prim, use_bg):
    rng = np.random.default_rng(5 + prim)
    H, W, n = 96, 128, 400
    B = 16384 // (12 * 16) + 1
    p, nrm, col = _surfels(rng, n, 0.8, 3.0, 4.0)
    K = K_for(H, W)
    uvw = (K @ p.T).T
    uv = np.clip(uvw[:, :2] / (uvw[:, 2:3] + 1e-7), -1, [W, H]).astype(np.float32)
    zn = np.array([np.linalg.norm(p[:, 2])], np.float32)
    bg = rng.uniform(0, 1, (1, 3, H, W)).astype(np.float32) if use_bg else None
    bgl = np.array([-0.3], np.float32) if use_bg else None
    Kt, Ki = T(K).view(1, 9), T(np.linalg.inv(K).astype(np.float32)).view(1, 9)
    diam = 0.02 if prim == 1 else 0.025
    args = [Kt, Ki, T(p)[None], T(nrm)[None], T(col)[None]]
    opt = dict(uv=T(uv)[None], znorm=T(zn), bg=None if bg is None else T(bg), bg_logit=None if bgl is None else T(bgl))
    one = _splat_any(prim, *args, W, H, 1, diam=diam, **opt)
    rep = lambda t: None if t is None else t.expand(B, *t.shape[1:]).contiguous()
    many = _splat_any(prim, *[rep(t) for t in args], W, H, B, diam=diam, **{k: rep(v) for k, v in opt.items()})
    assert float(one[1].sum()) > 50
    for x, y in zip(many, one):
        for b in (0, B - 1):
            assert torch.equal(x[b], y[0]), (prim, use_bg, b)


# ---- r05: the screen boxes are conservative (what a tighter box must never break) -----------------------------------------------------------

@pytest.mark.parametrize("case", ["centred", "cropped_far_principal_point", "near_big_discs", "far_small_discs", "grazing_and_behind"])
def test_disc_screen_boxes_never_cut_a_covered_pixel(case):
    """Every pixel a surfel covers must lie inside its screen box.  K with a skew entry of 1e-30 renders the same bits (1e-30 * y vanishes in
    every float sum) but is 'non-standard' to disc_bbox, which then hands out WHOLE-IMAGE boxes: the boxed render (forward images, aux,
    backward gradients) must equal that box-free render up to the order of its float sums, with the identical coverage pattern -- over surfels at all depths, off-axis principal points (the reference
    pipeline's crops, utils/refinement.py:586-609), discs from sub-pixel to a third of the image, surfels behind the camera."""
    rng = np.random.default_rng(sum(map(ord, case)))
    H, W, n = 96, 128, 1500
    f = 45.0 * H / 32.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    if case == "centred":
        p, nrm, col = _surfels(rng, n, 1.2, 2.5, 4.5)
    elif case == "cropped_far_principal_point":
        K = np.array([[857.0, 0, -128.3], [0, 861.0, 7.2], [0, 0, 1]], np.float32)       # G14's regime: principal point outside the crop, fx != fy
        p, nrm, col = _surfels(rng, n, 1.0, 10.0, 14.0)
        p[:, 0] += 2.5
    elif case == "near_big_discs":
        p, nrm, col = _surfels(rng, 300, 0.2, 0.3, 0.8)                                   # disc radius 0.04 at z = 0.3: 18 px and more
    elif case == "far_small_discs":
        p, nrm, col = _surfels(rng, n, 6.0, 20.0, 40.0)                                   # sub-pixel discs
    else:
        p, nrm, col = _surfels(rng, n, 1.5, -0.5, 1.0)                                    # some behind the camera plane, some within rho of it
        nrm = rng.standard_normal((p.shape[0], 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)                                 # every orientation: grazing planes (|n.ray| < 0.01) among them
    Ks = K.copy(); Ks[0, 1] = 1e-30
    Ki, Kis = np.linalg.inv(K).astype(np.float32), np.linalg.inv(K).astype(np.float32)    # the same rays for both
    Lh = _lib.lib()
    outs = []
    pt, nt, ct, Kit = T(p)[None].contiguous(), T(nrm)[None].contiguous(), T(col)[None].contiguous(), T(Ki).view(1, 9)   # (kept alive: raw pointers below)
    for Kx in (K, Ks):
        Kt = T(Kx).view(1, 9)
        o = _splat(0, Kt, Kit, pt, nt, ct, W, H)
        color, mask, depth, nimg, aux, ws = o
        nb = p.shape[0]
        boxes = N(ws)[:nb * 4].reshape(nb, 4)
        g = [torch.ones_like(color), torch.ones_like(mask), torch.ones_like(depth), torch.ones_like(nimg)]
        gp, gn, ga = (torch.zeros(1, nb, 3, device=DEV) for _ in range(3))
        _lib.check(Lh.sdfr_splat_backward(0, _lib.ptr(Kt), _lib.ptr(Kit), _lib.ptr(pt), _lib.ptr(nt),
                                          _lib.ptr(ct), None, None, None, None, 1, nb, None, W, H, 0.04, 150.0, _lib.ptr(aux), _lib.ptr(color),
                                          _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg), _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(g[2]), _lib.ptr(g[3]),
                                          _lib.ptr(gp), _lib.ptr(gn), _lib.ptr(ga), _lib.stream_ptr()), "sdfr_splat_backward")
        outs.append((color, mask, depth, nimg, aux, gp, gn, ga, boxes))
    a, b = outs
    whole = (b[8][:, 0] == 0) & (b[8][:, 2] == W - 1) & (b[8][:, 1] == 0) & (b[8][:, 3] == H - 1)
    assert whole.all(), "the skewed K must switch the boxes off"
    area = ((a[8][:, 2] - a[8][:, 0] + 1).clip(0) * (a[8][:, 3] - a[8][:, 1] + 1).clip(0)).astype(np.int64)
    assert area.sum() < 0.7 * whole.sum() * W * H or case in ("near_big_discs", "grazing_and_behind")     # ... and the plain K must use them
    # equal up to the ORDER of the float sums (the forward groups a tile's candidates into shares of its own list, the backward sums a surfel's
    # pixels in the order of its own box): a cut pixel would show as a missing O(1) weight, not as a rounding difference
    assert torch.equal(a[1] > 0, b[1] > 0), "coverage pattern"
    for x, y, name in zip(a[:8], b[:8], ("color", "mask", "depth", "normals", "aux", "g_p", "g_n", "g_attr")):
        if name == "aux":
            x, y = x[..., :3], y[..., :3]                      # (nu, max logit, denominator; the 4th word is a bit field)
        xn, yn = N(x), N(y)
        sane = np.isfinite(xn) & np.isfinite(yn) & (np.abs(yn) < 1e20)        # (surfels within rho of the camera plane: sums that overflow in either order)
        scale = max(1.0, float(np.abs(yn[sane]).max())) if sane.any() else 1.0
        tol = (2e-4 if name.startswith("g_") else 2e-5) * scale
        assert sane.mean() > 0.99 and np.allclose(xn[sane], yn[sane], rtol=2e-4, atol=tol), (name, float(np.abs(xn[sane] - yn[sane]).max()), scale)
    assert float(a[1].sum()) > 20
