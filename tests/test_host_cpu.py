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
    dec, L = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
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
    dec, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
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


def test_c_abi_error_conventions_without_gpu():
    """argument validation happens before any HIP call: negative status + a message from sdfr_last_error(); no compute without a GPU."""
    import ctypes
    h = _lib.lib()
    assert h.sdfr_mlp_forward(None, None, 10, None, None, None) == -1
    assert b"NULL" in h.sdfr_last_error()
    assert h.sdfr_band_select(None, 10, 1, 0.03, None, 10, None, None, None, None) == -1
    assert h.sdfr_splat_forward(7, None, None, None, None, None, None, None, None, None, 1, 0, None, 8, 8, 0.04, 150.0, None, None, None,
                                None, None, None, None) == -1
    assert b"primitive" in h.sdfr_last_error()
    hd = ctypes.c_void_p()
    ints = (ctypes.c_int * 2)(6, 64)
    outs = (ctypes.c_int * 2)(64, 2)                     # last layer must have out_dim 1
    z = (ctypes.c_int * 2)(0, 0)
    W = (ctypes.c_void_p * 2)(None, None)
    assert h.sdfr_decoder_create(ctypes.byref(hd), 2, ints, outs, z, z, W, W, None, None, 6, 0, 0) == -1
    assert b"out_dim 1" in h.sdfr_last_error()
    assert h.sdfr_decoder_mask_words(None, 100) == 0 and h.sdfr_decoder_macs(None) == 0
    with pytest.raises(_lib.SdfrError):
        _lib.check(-1, "demo")


def test_ctypes_prototypes_match_the_header():
    """every prototype in include/sdfr.h has a ctypes binding with the same number of parameters and a compatible kind per slot
    (pointer vs integer vs float); the .hip sources include the same header, so definitions cannot drift from it either."""
    import ctypes
    header = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|const char\*)\s+(sdfr_[a-z0-9_]+)\s*\((.*?)\)\s*;", header, flags=re.S)
    assert len(protos) == len(_lib.EXPORTS)
    for name, params in protos:
        params = [p.strip() for p in params.replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]
        res, args = _lib._PROTOS[name]
        assert len(params) == len(args), (name, len(params), len(args))
        for p, a in zip(params, args):
            if "*" in p:
                assert a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "contents") or a.__name__.startswith("LP_"), (name, p, a)
            elif p.startswith("float"):
                assert a is ctypes.c_float, (name, p)
            elif p.startswith("int64_t"):
                assert a is ctypes.c_int64, (name, p)
            else:
                assert a is ctypes.c_int, (name, p, a)


def test_optimizer_mirror_constructor_and_refusals():
    """pipelines/optimizer.py:26-54: params become float32 leaf tensors in place; what is off the path is refused loudly"""
    import pytest
    from sdflabel_amd.pipelines.optimizer import Optimizer, get_opt_params
    params = {"yaw": np.array([0.5]), "trans": np.array([0.1, 0.2, 3.0]), "scale": np.array([2.0]), "latent": np.zeros(3)}
    opt = Optimizer(params, "cpu", {"2d": 0.3, "3d": 0.5})
    assert opt.params is params
    for k, n in (("yaw", 1), ("trans", 3), ("scale", 1), ("latent", 3)):
        assert torch.is_tensor(params[k]) and params[k].dtype == torch.float32 and params[k].requires_grad and params[k].numel() == n
    assert [g["lr"] for g in opt.optim_params] == [0.01, 0.01, 0.01, 0.00003]
    with pytest.raises(NotImplementedError):
        Optimizer(dict(params), "cpu", {}, rot="quat")
    with pytest.raises(NotImplementedError):
        opt.optimize(1, None, np.zeros((1, 3)), None, None, None, (8, 8), viz_type="2d")


def test_c_abi_compiles_and_links_from_plain_c(tmp_path):
    """include/sdfr.h is a C header (no C++, no torch types): a C99 client compiles against it and links with the in-tree library"""
    from tests._util import build_c_abi_smoke
    exe = build_c_abi_smoke(tmp_path)
    assert os.path.isfile(exe)


@pytest.mark.parametrize("compiler,std,ext", [("gcc", "-std=c99", ".c"), ("g++", "-std=c++11", ".cpp")])
def test_public_header_is_strict_c_and_cxx(tmp_path, compiler, std, ext):
    """include/sdfr.h alone, with -Wall -Wextra -pedantic -Werror: plain pointers and sizes only, usable from C and from C++"""
    import subprocess
    src = tmp_path / ("hdr" + ext)
    src.write_text('#include "sdfr.h"\nint main(void) { return sdfr_version() > 0 ? 0 : 0; }\n')
    r = subprocess.run([compiler, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]


def test_sphere_tracer_default_schedule_is_a_function_of_the_crop_size_alone():
    """host logic of the sphere-tracing mode: the first speculative pass of the default schedule (measured optima per crop size)"""
    from sdflabel_amd.renderer.sphere_tracer import default_spec_from
    assert [default_spec_from(n * n, True) for n in (64, 128, 256, 512)] == [6, 8, 10, 13]
    assert [default_spec_from(n * n, False) for n in (64, 128, 256, 512)] == [6, 8, 10, 18]
    assert default_spec_from(1, True) == 4 and default_spec_from(1 << 30, False) == 24
    # behind the cone phase (r04 default) the float16 march starts its speculative passes earlier
    assert [default_spec_from(n * n, True, True) for n in (64, 128, 256, 512)] == [3, 4, 6, 8]
    assert [default_spec_from(n * n, False, True) for n in (128, 256)] == [8, 10]
    from sdflabel_amd.renderer.sphere_tracer import default_q_max, default_spec_levels
    assert default_spec_levels(256 * 256, True, True) == [(6, 4), (10, 16)] and default_spec_levels(512 * 512, True, True) == [(8, 4), (11, 16)]
    assert default_spec_levels(256 * 256, False, False) == [(10, 4), (13, 16)] and default_q_max() == 1.5


def test_no_memset_nodes_in_the_library():
    """every entry point may be captured into a HIP graph, and memset nodes faulted on replay (r04): the kernels' sources zero memory with
    sdfr_zero_async only"""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdflabel_amd", "csrc")
    for path in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        src = "\n".join(l.split("//")[0] for l in open(path).read().splitlines())
        assert "hipMemset" not in src, path


# ---- r05: the printed bench line (a 24 KB line could not be parsed by the driver) -----------------------------------------------

def _bench_module():
    import importlib
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_line_is_compact_and_round_trips():
    import json
    B = _bench_module()
    line = B.compact_line(B.CANNED_LINE, "bench_extras.json")
    assert "\n" not in line and len(line.encode()) < 4096
    d = json.loads(line)
    assert d["value"] == pytest.approx(B.CANNED_LINE["value"], rel=1e-6) and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] == pytest.approx(0.871136)
    assert d["cpu_baseline"]["cores"] == 8 and d["cpu_baseline"]["kind"] == "port" and len(d["cpu_baseline"]["sample"]) <= 240
    assert d["refine"] == {"f32_crops_per_s": pytest.approx(9.508298), "f16_crops_per_s": pytest.approx(71.89664), "total_crops": 1024,
                           "iterations_per_crop": 60, "f16_candidate_reuse": True, "f16_full_grid_passes_per_crop": pytest.approx(1.0),
                           "area32_f16_crops_per_s": pytest.approx(301.5)}
    assert d["extras"] == "bench_extras.json"


def test_bench_line_key_set_is_pinned():
    """a new section belongs in bench_extras.json, not on the line: adding a top-level key fails here first"""
    import json
    B = _bench_module()
    d = json.loads(B.compact_line(dict(B.CANNED_LINE, a_new_section={"x": 1}), None))
    assert tuple(d.keys()) == ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "refine", "extras") == B.LINE_KEYS
    assert set(d["roofline"]) <= set(B.ROOFLINE_KEYS) and set(d["config"]) <= set(B.CONFIG_KEYS) and set(d["cpu_baseline"]) <= set(B.CPU_KEYS)
    # a multi-rank run has no cpu_baseline (rank 0 at N = 1 only): null, not missing
    assert json.loads(B.compact_line({k: v for k, v in B.CANNED_LINE.items() if k != "cpu_baseline"}, None))["cpu_baseline"] is None


def test_bench_line_refuses_to_grow_past_the_bound():
    B = _bench_module()
    fat = dict(B.CANNED_LINE, config=dict(B.CANNED_LINE["config"], **{k: "q" * 199 for k in B.CONFIG_KEYS}),
               roofline={k: "r" * 119 for k in B.ROOFLINE_KEYS}, cpu_baseline={k: "s" * 239 for k in B.CPU_KEYS}, metric="m" * 3000)
    with pytest.raises(RuntimeError, match="bench line"):
        B.compact_line(fat, None)


def test_bench_extras_file_holds_every_section(tmp_path):
    import json
    B = _bench_module()
    p = str(tmp_path / "extras.json")
    B.write_extras(B.CANNED_LINE, p)
    assert json.load(open(p)).keys() == B.CANNED_LINE.keys()


# ---- r05: the build is locked and the ABI version is checked ------------------------------------------------

def test_in_tree_library_is_a_product_build_of_this_header():
    h = sdflabel_amd.lib()
    assert h.sdfr_build_flags() == 0
    hdr = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    assert int(re.search(r"#define SDFR_VERSION (\d+)", hdr).group(1)) == h.sdfr_version() == _lib.ABI_VERSION


def test_build_script_ignores_define_hooks_unless_asked_for_an_experiment_build():
    import subprocess
    sh = os.path.join(ROOT, "sdflabel_amd", "csrc", "build.sh")
    env = dict(os.environ, SDFR_BUILD_DRYRUN="1", SDFR_F16_DEFS="-DSDFR_ABL_NOMFMA", SDFR_FWD_DEFS="-DSDFR_FWD_PF=3")
    env.pop("SDFR_AB", None)
    r = subprocess.run(["bash", sh], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "defs:[]" in r.stdout and "SDFR_EXPERIMENT" not in r.stdout and "ignoring SDFR_F16_DEFS" in r.stderr
    r = subprocess.run(["bash", sh], capture_output=True, text=True, env=dict(env, SDFR_AB="1"))
    assert r.returncode == 0 and "-DSDFR_ABL_NOMFMA" in r.stdout and "-DSDFR_EXPERIMENT=1" in r.stdout
    # a product build cannot be redirected either
    r = subprocess.run(["bash", sh], capture_output=True, text=True, env=dict(env, SDFR_LIBNAME="libx.so"))
    assert r.returncode == 2


def test_binding_refuses_an_experiment_library_and_a_foreign_abi(monkeypatch):
    class Fake:
        def __init__(self, version, flags):
            self._v, self._f = version, flags

        def __getattr__(self, name):
            if name == "sdfr_version":
                return lambda: self._v
            if name == "sdfr_build_flags":
                return lambda: self._f
            return type("F", (), {"restype": None, "argtypes": None})()

    import ctypes
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(ctypes, "CDLL", lambda p: Fake(_lib.ABI_VERSION, 1))
    monkeypatch.delenv("SDFR_ALLOW_AB", raising=False)
    with pytest.raises(_lib.SdfrError, match="experiment build"):
        _lib.lib()
    monkeypatch.setattr(ctypes, "CDLL", lambda p: Fake(200, 0))
    with pytest.raises(_lib.SdfrError, match="ABI version 200"):
        _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)


# ---- r05: the proven Lipschitz bound behind the candidate reuse -------------------------------------------------------------------------------

@pytest.mark.parametrize("kw", [None, dict(latent_in=[2, 4], xyz_in_all=False), dict(latent_in=[3], xyz_in_all=True), dict(latent_in=(), xyz_in_all=False)])
def test_latent_lipschitz_bound_is_an_upper_bound(kw):
    """Decoder.latent_lipschitz_bound() >= every sampled |sdf(z + dz, x) - sdf(z, x)| / |dz| (oracle forward, float64 differences), for the
    shipped decoder and random decoders with other injection patterns; inf for LayerNorm decoders"""
    rng = np.random.default_rng(5)
    if kw is None:
        d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
        st, spec = fitted_state()
        layers = O.decoder_layers_from_state(st, spec)
        assert 100.0 < d.latent_lipschitz_bound() < 2000.0
    else:
        torch.manual_seed(3)
        dims = [64, 64, 64, 64, 64, 64]
        d = sdflabel_amd.Decoder(3, dims=dims, norm_layers=(), weight_norm=False, **kw).eval()
        with torch.no_grad():
            for p in d.parameters():
                p.mul_(1.7)
        layers = [(W, b, None) for W, b in d.effective_layers()]
        spec = dict(dims=dims, latent_in=list(kw["latent_in"]), xyz_in_all=kw["xyz_in_all"])
    bound = d.latent_lipschitz_bound()
    worst = 0.0
    for _ in range(24):
        z = rng.normal(size=3); z /= np.linalg.norm(z)
        dz = rng.normal(size=3) * 10.0 ** rng.uniform(-4, -1)
        x = rng.uniform(-1, 1, size=(512, 3))
        a = np.concatenate([np.tile(z, (512, 1)), x], 1).astype(np.float32)
        b = a.copy(); b[:, :3] = (z + dz).astype(np.float32)
        step = np.linalg.norm(b[0, :3].astype(np.float64) - a[0, :3].astype(np.float64))
        fa, fb = O.decoder_forward(layers, spec, a).astype(np.float64), O.decoder_forward(layers, spec, b).astype(np.float64)
        worst = max(worst, float(np.abs(fa - fb).max() / step))
    assert 0.0 < worst <= bound, (worst, bound)
    if kw is not None and not kw["latent_in"]:
        ln = sdflabel_amd.Decoder(3, dims=[32, 32], norm_layers=(0, 1), weight_norm=False)
        assert ln.latent_lipschitz_bound() == float("inf")


def test_stored_traffic_is_quoted_only_for_the_kernel_sources_it_was_measured_on(tmp_path):
    """profiles/traffic_*.json carry the hash of the csrc files of their kernels; bench.py quotes the figure only when the hash
    matches this tree (otherwise roofline.traffic is null instead of a number measured on other code)"""
    import json
    from sdflabel_amd import _lib
    name = "traffic_mlp_forward.json"
    os.makedirs(tmp_path / "profiles")
    good = {"hbm_bytes_per_launch": 1.0, "source_sha16": _lib.source_sha16(_lib.TRAFFIC_SOURCES[name])}
    json.dump(good, open(tmp_path / "profiles" / name, "w"))
    assert _lib.stored_traffic(str(tmp_path), name)["hbm_bytes_per_launch"] == 1.0
    json.dump(dict(good, source_sha16="0" * 16), open(tmp_path / "profiles" / name, "w"))
    assert _lib.stored_traffic(str(tmp_path), name) is None
    json.dump({"hbm_bytes_per_launch": 1.0}, open(tmp_path / "profiles" / name, "w"))          # a file from before r06: no hash
    assert _lib.stored_traffic(str(tmp_path), name) is None
    assert _lib.stored_traffic(str(tmp_path), "traffic_splat.json") is None                    # absent
    assert len(_lib.source_sha16(("splat.hip",))) == 16 and _lib.source_sha16(("splat.hip",)) != _lib.source_sha16(("trace.hip",))
