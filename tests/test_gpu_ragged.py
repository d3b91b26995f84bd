"""GPU tests of ragged extents: every crop of a batch its own image size (H_b, W_b) and intrinsics K_b, as the crops of the
reference pipeline have (utils/refinement.py:586-609 adjust_intrinsics_crop; pipelines/refine_css.py:117-129,203-223) -- read by the kernels from
device memory, so one BatchRenderer / BatchRefiner (one set of buffers, one captured HIP graph) serves any crop set within its capacity.
  * G14's three crops (160x306, 155x316, 156x313 rays; principal points outside the crops; fx != fy) rendered in ONE ragged batch: images,
    surfels and gradients against the goldens AND bit-identical to each crop rendered alone at its own size;
  * G14o's crop (23x44 rays, the reference Optimizer's own trajectory) refined in one ragged batch with crops of other sizes: the golden
    trajectory, and every crop bit-identical to the same crop refined alone by a fixed-size refiner; the captured graph survives new crop sets;
  * the product-side Optimizer shares ONE refiner (and graph) across crops of different sizes and intrinsics."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from sdflabel_amd.fixtures import GT_LATENT, GT_SCALE, GT_TRANS, GT_YAW
from tests._util import ASSET, K_for, gold, pattern_weights
from tests.test_gpu_configs import SALT, check_images
from tests.test_gpu_cropped import Sub, _weights, grads_close
from tests.test_gpu_parity import N, T

pytestmark = pytest.mark.gpu
DEV = "cuda"
WEIGHTS = {"2d": 0.3, "3d": 0.5}


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


def test_g14_three_crops_of_different_sizes_in_one_batch(dec):
    zs = [Sub(gold("g14_cropped_intrinsics.npz"), t + "_") for t in ("a", "b", "c")]
    cfgs = [[int(v) for v in z["cfg"]] for z in zs]
    D = cfgs[0][0]
    assert all(c[0] == D for c in cfgs) and len({(c[1], c[2]) for c in cfgs}) == 3
    sizes = [(c[2], c[1]) for c in cfgs]                                              # (W_b, H_b)
    PS = max(w * h for w, h in sizes)
    B = 3
    br = sdflabel_amd.BatchRenderer(dec, D, zs[0]["K"], sizes[0], B, device=DEV, max_pixels=PS, max_side=512)
    br.set_extents(sizes, np.stack([z["K"] for z in zs]))
    yaw = T(np.concatenate([z["yaw"] for z in zs]))
    trans = T(np.stack([z["trans"] for z in zs]))
    lat = T(np.stack([z["latent"] for z in zs]))
    out = br.forward(yaw, trans, lat)
    assert not br.overflow() and br.ragged
    g_img = {k: torch.zeros_like(out[k]) for k in ("color", "mask", "depth", "normals")}
    gx = torch.zeros(B, br.cap, 3, device=DEV)
    alone = []
    for b, z in enumerate(zs):
        _, H, W = cfgs[b]
        near = np.unpackbits(z["near_threshold"])[:H * W].astype(bool)
        nf = z["xyzf"].shape[0]
        assert int(out["n"][b]) == z["pcd"].shape[0] and int(out["nf"][b]) == nf
        imgs = {k: br.image(b, k) for k in ("color", "mask", "depth", "normals")}
        check_images(imgs, z, near=near)
        assert np.abs(N(out["xyzf"][b, :nf]) - z["xyzf"]).max() < 1e-5
        _, wr = _weights(z, imgs, near)
        for k in g_img:
            g_img[k][b, :, :H * W] = wr[k].reshape(wr[k].shape[0], H * W)
        gx[b, :nf] = T(pattern_weights((nf, 3), SALT["xyzf"]))
        # the same crop alone, fixed extents
        one = sdflabel_amd.BatchRenderer(dec, D, z["K"], (W, H), 1, device=DEV)
        o1 = one.forward(yaw[b:b + 1], trans[b:b + 1], lat[b:b + 1])
        for k in imgs:
            assert torch.equal(o1[k][0], imgs[k]), (b, k)
        assert torch.equal(o1["xyzf"][0], out["xyzf"][b]) and torch.equal(one.aux[0, :H * W], br.aux[b, :H * W])
        g1 = one.backward(g_color=wr["color"][None], g_mask=wr["mask"][None], g_depth=wr["depth"][None], g_normals=wr["normals"][None], g_xyzf=gx[b:b + 1])
        alone.append([t.clone() for t in g1])
    g = br.backward(g_color=g_img["color"], g_mask=g_img["mask"], g_depth=g_img["depth"], g_normals=g_img["normals"], g_xyzf=gx)
    for b, z in enumerate(zs):
        grads_close([t[b] for t in g], z, "r_g_", 1e-3)                               # against the reference's autograd gradients
        for x, y in zip(alone[b], g):
            assert torch.equal(x[0], y[b]), b                                         # and bit-identical to the crop alone
    # a different crop set through the same buffers: sizes permuted
    perm = [2, 0, 1]
    br.set_extents([sizes[i] for i in perm], np.stack([zs[i]["K"] for i in perm]))
    o2 = br.forward(yaw[perm], trans[perm], lat[perm])
    for b, i in enumerate(perm):
        _, H, W = cfgs[i]
        check_images({k: br.image(b, k) for k in ("color", "mask", "depth", "normals")}, zs[i], near=np.unpackbits(zs[i]["near_threshold"])[:H * W].astype(bool))
    with pytest.raises(sdflabel_amd.SdfrError):
        br.set_extents([(600, 90)] * 3)                                               # beyond max_side / max_pixels: refused, not truncated


def _synthetic_crop(dec, D, H, W, K):
    """target NOCS image and lidar-like cloud of the ground-truth pose at this crop's size and intrinsics (fixtures.synthetic_targets with a K)"""
    gt = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=DEV)
    o = gt.forward(torch.tensor([GT_YAW], device=DEV), torch.tensor([GT_TRANS], device=DEV), torch.tensor([GT_LATENT], device=DEV))
    nf = int(o["nf"][0])
    return o["color"][0].clone(), (o["xyzf"][0, :nf] * GT_SCALE)[::2].cpu().numpy()


def test_g14o_and_other_crop_sizes_refined_in_one_ragged_batch(dec):
    z = Sub(gold("g14o_optimizer_cropped.npz"), "a_")
    D, H0, W0 = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    # crop 0: the golden's (23x44, cropped intrinsics, the reference's own targets); crops 1..3: other sizes / aspects with centred intrinsics
    shapes = [(H0, W0), (30, 34), (18, 56), (40, 24)]
    Ks = [z["K"]] + [K_for(h, w) for h, w in shapes[1:]]
    targets, lidars = [T(z["nocs_target"])], [z["lidar"]]
    for (h, w), K in zip(shapes[1:], Ks[1:]):
        tg, ld = _synthetic_crop(dec, D, h, w, K)
        targets.append(tg); lidars.append(ld)
    B = len(shapes)
    rng = np.random.default_rng(4)
    par = {"yaw": np.concatenate([init[0:1], GT_YAW + rng.uniform(0.05, 0.15, B - 1)]).astype(np.float32),
           "trans": np.concatenate([init[None, 1:4], np.asarray(GT_TRANS)[None] + rng.uniform(-0.05, 0.05, (B - 1, 3))]).astype(np.float32),
           "scale": np.concatenate([init[4:5], np.full(B - 1, GT_SCALE)]).astype(np.float32),
           "latent": np.concatenate([init[None, 5:8], np.asarray(GT_LATENT)[None] + rng.uniform(-0.1, 0.1, (B - 1, 3))]).astype(np.float32)}
    lcap = 1 << (max(l.shape[0] for l in lidars) - 1).bit_length()
    rf = sdflabel_amd.BatchRefiner(dec, D, Ks[0], shapes[0], B, lidar_cap=lcap, weights=WEIGHTS, device=DEV, max_pixels=2048, max_side=128)
    rf.set_crops(par, targets, lidars, K=np.stack(Ks), crop_sizes=shapes)
    rf.capture()
    traj = []
    for _ in range(10):
        rf.optimize(1)
        traj.append(N(rf.results()[0]))
        assert int(rf.stepped.min()) == 1
    traj = np.asarray(traj)
    assert np.abs(traj[:, 0] - z["traj"]).max() < 5e-4, np.abs(traj[:, 0] - z["traj"]).max(axis=0)      # the reference Optimizer's own trajectory
    l2, l3 = N(rf.results()[1]), N(rf.results()[2])
    assert abs(l2[0] - z["loss2d_weighted"][-1]) < 2e-4 and abs(l3[0] - z["loss3d_weighted"][-1]) < 2e-4
    # every crop alone in a fixed-size refiner: the same bits
    for b, (h, w) in enumerate(shapes):
        one = sdflabel_amd.BatchRefiner(dec, D, Ks[b], (h, w), 1, lidar_cap=lcap, weights=WEIGHTS, device=DEV)
        one.set_crops({k: v[b:b + 1] for k, v in par.items()}, targets[b][None], [lidars[b]])
        one.capture()
        one.optimize(10)
        assert np.array_equal(N(one.results()[0])[0], traj[-1, b]), (b, N(one.results()[0])[0], traj[-1, b])
    # a new crop set (sizes rotated) through the SAME captured graph
    rot = [1, 2, 3, 0]
    rf.set_crops({k: v[rot] for k, v in par.items()}, [targets[i] for i in rot], [lidars[i] for i in rot], K=np.stack([Ks[i] for i in rot]),
                 crop_sizes=[shapes[i] for i in rot])
    rf.optimize(10)
    assert rf.captures == 1                                                                                 # no re-capture for the new sizes
    rows = N(rf.results()[0])
    for b, i in enumerate(rot):
        assert np.array_equal(rows[b], traj[-1, i]), (b, i)


def test_optimizer_mirror_shares_one_refiner_across_crop_sizes(dec):
    """the reference's callers construct an Optimizer per annotation with that crop's own size and K (refine_css.py:203-223): the product-side
    mirror serves them from ONE cached refiner (no re-allocation, no re-capture) and reproduces the golden trajectory end state"""
    from sdflabel_amd.pipelines import optimizer as OP
    OP.clear_refiner_cache()
    za, zc = Sub(gold("g14o_optimizer_cropped.npz"), "a_"), Sub(gold("g14o_optimizer_cropped.npz"), "c_")
    D = int(za["D"])
    grid = sdflabel_amd.Grid3D(D, DEV)
    ends = []
    seen = set()
    for it, (H, W, K, tgt, lid, init) in enumerate([(int(za["H"]), int(za["W"]), za["K"], za["nocs_target"], za["lidar"], za["init"]),
                                                     (30, 34, K_for(30, 34), None, None, None), (int(za["H"]), int(za["W"]), za["K"], za["nocs_target"], za["lidar"], za["init"])]):
        if tgt is None:
            t_, l_ = _synthetic_crop(dec, D, H, W, K)
            tgt, lid = N(t_), l_
            init = np.array([GT_YAW + 0.1, GT_TRANS[0] + 0.03, GT_TRANS[1], GT_TRANS[2] - 0.05, GT_SCALE, *GT_LATENT], np.float32)
        p = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
        opt = OP.Optimizer(p, DEV, WEIGHTS)
        out = opt.optimize(10, torch.from_numpy(np.asarray(tgt)), lid, dec, grid, torch.from_numpy(np.asarray(K, np.float32)), [H, W])
        seen.add(id(opt._refiner))
        ends.append(np.concatenate([N(out[k]).ravel() for k in ("yaw", "trans", "scale", "latent")]))
    assert len(seen) == 1 and len(OP._REFINERS) == 1                                   # one refiner, one graph, three crops of two sizes
    assert np.abs(ends[0] - za["traj"][-1]).max() < 5e-4
    assert np.array_equal(ends[0], ends[2])                                            # the refiner carries nothing over from the crop in between
    OP.clear_refiner_cache()


def test_degenerate_extents_in_one_ragged_batch(dec):
    """edge cases of the extents: a 1x1 crop, one-pixel-wide strips in both directions, a crop that fills the pixel capacity exactly and a tiny odd one,
    side by side in one batch -- splat path: every image and gradient bit-identical to the same crop rendered alone at its own fixed size; tracer:
    identical to the same crop alone in a ragged tracer of the same capacity, no hit outside a crop's own pixels; nothing reads or writes out of bounds
    (the box would fault) and every value is finite."""
    D = 40
    sizes = [(1, 1), (1, 64), (64, 1), (64, 64), (3, 5), (7, 2)]                     # (W_b, H_b)
    B, PS = len(sizes), 64 * 64
    Ks = []
    for w, h in sizes:
        K_ = K_for(max(h, 8), max(w, 8))                                              # a focal length that keeps the object in a tiny crop's view
        K_[0, 2], K_[1, 2] = w / 2.0, h / 2.0
        Ks.append(K_)
    Ks = np.stack(Ks)
    yaw = T(np.linspace(0.2, 0.9, B).astype(np.float32))
    trans = T(np.tile(np.array([[0.0, 0.0, 3.5]], np.float32), (B, 1)))
    lat = T(np.tile(np.array([[0.3, -0.5, 0.8]], np.float32), (B, 1)))
    br = sdflabel_amd.BatchRenderer(dec, D, Ks[0], sizes[0], B, device=DEV, max_pixels=PS, max_side=64)
    br.set_extents(sizes, Ks)
    out = br.forward(yaw, trans, lat)
    g_img = {k: torch.zeros_like(out[k]) for k in ("color", "mask", "depth", "normals")}
    for b, (w, h) in enumerate(sizes):
        for k in g_img:
            g_img[k][b, :, :w * h] = T(pattern_weights((g_img[k].shape[1], w * h), SALT[k]))
    g = [t.clone() for t in br.backward(g_color=g_img["color"], g_mask=g_img["mask"], g_depth=g_img["depth"], g_normals=g_img["normals"])]
    covered = 0
    for b, (w, h) in enumerate(sizes):
        one = sdflabel_amd.BatchRenderer(dec, D, Ks[b], (w, h), 1, device=DEV)
        o1 = one.forward(yaw[b:b + 1], trans[b:b + 1], lat[b:b + 1])
        for k in g_img:
            img = br.image(b, k)
            assert img.shape[1:] == (h, w) and bool(torch.isfinite(img).all()) and torch.equal(o1[k][0], img), (b, k)
            assert float(out[k][b, :, w * h:].abs().sum()) == 0.0, (b, k)             # nothing beyond the crop's own pixels
        g1 = one.backward(**{"g_" + k: g_img[k][b:b + 1, :, :w * h].reshape(1, -1, h, w) for k in g_img})
        for x, y in zip(g1, g):
            assert bool(torch.isfinite(y[b]).all()) and torch.equal(x[0], y[b]), b
        covered += int((br.image(b, "mask") > 0).sum())
    assert covered > 1000                                                             # (the 64x64 crop sees the object; the strips may or may not)
    # the sphere tracer on the same extents
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    d16 = d16.to(DEV)
    tr = sdflabel_amd.SphereTracer(d16, Ks[0], sizes[0], B, device=DEV, max_pixels=PS, max_side=64, points=True)
    tr.set_extents(sizes, Ks)
    ot = {k: v.clone() for k, v in tr.render(yaw, trans, lat).items()}
    gt = [t.clone() for t in tr.backward(g_color=g_img["color"], g_depth=g_img["depth"], g_normals=g_img["normals"])]
    assert tr.stats()["unresolved"] == 0
    one = sdflabel_amd.SphereTracer(d16, Ks[0], sizes[0], 1, device=DEV, max_pixels=PS, max_side=64, points=True)
    for b, (w, h) in enumerate(sizes):
        one.set_extents([sizes[b]], Ks[b])
        o1 = one.render(yaw[b:b + 1], trans[b:b + 1], lat[b:b + 1])
        for k in ("color", "mask", "depth", "normals"):
            assert bool(torch.isfinite(ot[k][b]).all()) and torch.equal(o1[k][0], ot[k][b]), (b, k)
            assert float(ot[k][b, :, w * h:].abs().sum()) == 0.0, (b, k)
        nf = int(ot["nf"][b])
        assert int(o1["nf"][0]) == nf and torch.equal(o1["xyzf"][0, :nf], ot["xyzf"][b, :nf])       # (rows beyond the count are not written)
        g1 = one.backward(g_color=g_img["color"][b:b + 1], g_depth=g_img["depth"][b:b + 1], g_normals=g_img["normals"][b:b + 1])
        for x, y in zip(g1, gt):
            assert bool(torch.isfinite(y[b]).all()) and torch.equal(x[0], y[b]), b
    assert int(ot["nf"][3]) > 500


def test_extent_outside_the_contract_renders_as_an_empty_crop_not_out_of_bounds(dec):
    """the `_r` entry points read (W_b, H_b) from device memory and cannot validate them on the host.  An extent with more pixels than
    the slot (or a non-positive side), written straight to the device behind set_extents()'s back, must leave that crop EMPTY -- images
    untouched, zero gradients, nothing written outside its slot -- while the other crops of the batch render bit-identically to a clean run.
    Both renderers."""
    D, B, PS = 16, 3, 32 * 32
    K = K_for(32, 32)
    yaw = T(np.array([0.6, 0.2, -0.4], np.float32)); trans = T(np.array([[0.0, 0.0, 3.5]] * 3, np.float32))
    lat = T(np.array([[0.3, -0.5, 0.8]] * 3, np.float32))
    sizes = [(32, 32), (24, 40), (40, 20)]
    for bad in ((64, 64), (0, 16), (-3, 7)):
        br = sdflabel_amd.BatchRenderer(dec, D, K, (32, 32), B, device=DEV, max_pixels=PS, max_side=64)
        br.set_extents(sizes)
        out = br.forward(yaw, trans, lat)
        g = br.backward(g_color=torch.ones_like(br.color), g_mask=torch.ones_like(br.mask))
        clean = [t.clone() for t in (br.color, br.mask, br.depth, br.nimg, br.aux, g[0], g[1], g[2])]
        assert float(clean[1][1].sum()) > 5
        with torch.no_grad():
            br.wh[1] = torch.tensor(bad, dtype=torch.int32, device=DEV)         # behind the validator's back
        for t in (br.color, br.mask, br.depth, br.nimg):
            t.fill_(7.0)
        br.forward(yaw, trans, lat)
        g = br.backward(g_color=torch.ones_like(br.color), g_mask=torch.ones_like(br.mask))
        now = [br.color, br.mask, br.depth, br.nimg, br.aux, g[0], g[1], g[2]]
        for b in (0, 2):
            w, h = sizes[b]
            for a, c in zip(now[:4], clean[:4]):
                assert torch.equal(a[b, :, :w * h], c[b, :, :w * h]), (bad, b)
            for a, c in zip(now[5:], clean[5:]):
                assert torch.equal(a[b], c[b]), (bad, b)
        assert bool((br.color[1] == 7.0).all()) and bool((br.mask[1] == 7.0).all())      # the bad crop's slot: untouched
        assert float(g[0][1].abs()) == 0.0 and float(g[1][1].abs().max()) == 0.0
    # the tracer's readers
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    tr = sdflabel_amd.SphereTracer(d16.to(DEV), K, (32, 32), B, steps=32, device=DEV, max_pixels=PS, max_side=64)
    tr.set_extents(sizes)
    tr.render(yaw, trans, lat)
    clean = [t.clone() for t in (tr.color, tr.mask, tr.depth)]
    with torch.no_grad():
        tr.wh[1] = torch.tensor((64, 64), dtype=torch.int32, device=DEV)
    tr.render(yaw, trans, lat)
    for b in (0, 2):
        w, h = sizes[b]
        for a, c in zip((tr.color, tr.mask, tr.depth), clean):
            assert torch.equal(a[b, :, :w * h], c[b, :, :w * h]), b
    assert float(tr.mask[1].sum()) == 0.0 or bool((tr.mask[1] == clean[1][1]).all())       # nothing new rendered into the bad crop's slot
