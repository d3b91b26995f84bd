"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, host logic, loud failure without GPU."""
import os
import re

import numpy as np
import pytest
import torch

import sdflabel_amd
from sdflabel_amd import _lib
from tests._util import ROOT, gold, fitted_state, ASSET
from oracle import sdf_oracle as O


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    declared = set(re.findall(r"\b(sdfr_[a-z0-9_]+)\s*\(", header))
    declared -= {"sdfr_decoder"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name), name
    assert h.sdfr_version() >= 100


def test_grid_points_match_reference_golden():
    z = gold("g1_grid.npz")
    for D in (4, 5, 8):
        g = sdflabel_amd.Grid3D(D)
        assert g.points.requires_grad and g.points.is_leaf and g.points.dtype == torch.float32
        assert np.array_equal(g.points.detach().numpy(), z["grid_%d" % D])
    for D in (30, 40):
        g = sdflabel_amd.Grid3D(D).points.detach().numpy()
        assert np.array_equal(g[::97], z["grid_%d_stride97" % D])


def test_setup_dsdf_loads_reference_format_and_folds_weight_norm():
    dec, L = sdflabel_amd.setup_dsdf(ASSET + ".pt")
    assert L == 3 and not dec.training
    st, spec = fitted_state()
    ref_layers = O.decoder_layers_from_state(st, spec)
    ours = dec.effective_layers()
    assert len(ours) == len(ref_layers) == 9
    for (W, b), (Wr, br, _) in zip(ours, ref_layers):
        assert W.shape == Wr.shape
        assert np.allclose(W, Wr, atol=1e-7) and np.array_equal(b, br)
    assert dec._inject_table()[4] == (6, 0) and sum(i[0] for i in dec._inject_table()) == 6
    dec16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    assert dec16.mlp_precision == torch.float16 and next(dec16.parameters()).dtype == torch.float32
    with pytest.raises(NotImplementedError):
        sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.bfloat16)


def test_no_cpu_fallback():
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt")
    with pytest.raises(_lib.SdfrError):
        dec(torch.zeros(8, 6))
    g = sdflabel_amd.Grid3D(4)
    with pytest.raises(_lib.SdfrError):
        g.get_surface_points(torch.zeros(64, 1))
    r = sdflabel_amd.Rasterer(None, (16, 16))
    with pytest.raises(_lib.SdfrError):
        r(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 3), torch.eye(4), rot='dcm')


def test_rasterer_buffers():
    K = torch.tensor([[45., 0, 16], [0, 45., 16], [0, 0, 1]])
    r = sdflabel_amd.Rasterer(K, (32, 24))
    assert r.grid.shape == (1, 32 * 24, 2) and np.array_equal(r.grid[0].numpy(), O.pixel_grid((32, 24)))
    assert torch.allclose(r.Kinv @ r.K, torch.eye(3), atol=1e-6)
    r2 = sdflabel_amd.Rasterer(None, (200, 100))
    assert np.allclose(r2.K.numpy(), O.calibration_matrix((200, 100), 20, 70), rtol=1e-6)


def test_qrot_matrix_matches_qrot():
    from sdflabel_amd.renderer.utils_rasterer import qrot, qrot_matrix
    torch.manual_seed(0)
    q = torch.randn(4)
    v = torch.randn(7, 3)
    assert torch.allclose(qrot(q.expand(7, 4), v), v @ qrot_matrix(q).T, atol=1e-5)
    assert np.allclose(O.qrot(q.expand(7, 4).numpy(), v.numpy()), qrot(q.expand(7, 4), v).numpy(), atol=1e-5)
