"""GPU tests of the sphere-tracing render mode (not in the reference, so: self-consistency, agreement with the faithful
splat renderer up to the band thickness, gradients against finite differences)."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from tests._util import ASSET, K_for
from tests.test_gpu_parity import N, T

pytestmark = pytest.mark.gpu
DEV = "cuda"
YAW, TRANS, LAT = [0.6], [[0.05, -0.03, 3.5]], [[0.3, -0.5, 0.8]]


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


def _args(yaw=YAW, trans=TRANS, lat=LAT, grad=False):
    a = [torch.tensor(v, dtype=torch.float32, device=DEV) for v in (yaw, trans, lat)]
    return [t.requires_grad_(True) for t in a] if grad else a


def test_hits_lie_on_the_level_set_and_the_march_terminates(dec):
    H = W = 128
    st = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=96, device=DEV)
    out = st(*_args())
    assert st.n_hit > 2000 and st.n_entered > st.n_hit
    assert int(st.n_unresolved) <= 0.01 * st.n_entered                       # (grazing rays may still be creeping along the surface)
    assert float(st.hit_residual.abs().max()) < st.eps                       # the march stopped inside the tolerance
    # after the Newton polish the decoder vanishes at the hit points
    m = out["mask"][0, 0] > 0
    x = (out["color"][0].permute(1, 2, 0)[m] * 2 - 1) * torch.tensor([-1.0, 1.0, 1.0], device=DEV)      # NOCS colour -> object point
    latn = torch.nn.functional.normalize(torch.tensor(LAT[0], device=DEV), dim=0)
    sdf, _ = dec(torch.cat([latn.expand(x.shape[0], -1), x], 1).contiguous())
    r = N(sdf).reshape(-1)
    assert np.median(np.abs(r)) < 2e-5 and np.quantile(np.abs(r), 0.99) < 1e-3 and np.abs(r).max() < 2.5e-3, (np.median(np.abs(r)), np.quantile(np.abs(r), 0.99), np.abs(r).max())
    assert set(np.unique(N(out["mask"])).tolist()) == {0.0, 1.0}
    d = N(out["depth"][0, 0])[N(m)]
    assert d.min() > 2.0 and d.max() < 5.0
    nrm = N(out["normals"][0].permute(1, 2, 0)[m]) * 2 - 1
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-4 and (nrm[:, 2] < 0.2).mean() > 0.95     # camera-facing


def test_agrees_with_the_splat_renderer_up_to_the_band_thickness(dec):
    """the faithful path renders surfels of the |sdf| < 0.03 band of a 40^3 grid; the traced level set must give the same silhouette (up to the
    disc radius), the same depth (up to the band / disc size) and the same NOCS colours (both are object coordinates of surface points)"""
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

    functional(o0).backward()
    g = {"yaw": a0[0].grad, "trans": a0[1].grad, "latent": a0[2].grad}[which].view(-1)[index]
    fds = [float((functional(op) - functional(om)) / (2 * h)) for op, om, h in pairs]
    fd = float(np.mean(fds))
    assert abs(float(g) - fd) < 0.08 * max(1.0, abs(fd)), (float(g), fds)


def test_half_operand_march_and_batches(dec):
    """float16 decoder on the march (the hit polish stays exact f32): same image up to half precision; a batch renders each crop as alone"""
    H = W = 96
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    s32 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 1, steps=64, device=DEV)
    s16 = sdflabel_amd.SphereTracer(d16.to(DEV), K_for(H, W), (W, H), 1, steps=64, device=DEV)
    a, b = s32(*_args()), s16(*_args())
    assert s16.half == 1 and float((a["mask"] != b["mask"]).float().mean()) < 0.01
    both = (a["mask"] > 0) & (b["mask"] > 0)
    assert float(((a["depth"] - b["depth"]).abs() * both).max()) < 5e-3
    yaw, trans, lat = [0.6, -0.4], [[0.05, -0.03, 3.5], [0.1, 0.0, 3.0]], [[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]]
    s2 = sdflabel_amd.SphereTracer(dec, K_for(H, W), (W, H), 2, steps=64, device=DEV)
    o2 = s2(*_args(yaw, trans, lat))
    for i in range(2):
        o1 = s32(*_args(yaw[i:i + 1], trans[i:i + 1], lat[i:i + 1]))
        assert torch.equal(o1["mask"][0], o2["mask"][i])
        assert float((o1["depth"][0] - o2["depth"][i]).abs().max()) < 1e-5
