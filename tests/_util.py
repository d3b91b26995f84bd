"""Shared helpers for the tests (fixtures, decoder loading).  Never reads /root/reference."""
import json
import os

import numpy as np

from sdflabel_amd.fixtures import ASSET, K_for, fitted_state  # noqa: F401  (the fixtures live in the package: bench.py / tools/ use them too)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def state_from_npz(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def build_c_abi_smoke(out_dir):
    """Compile tests/c_abi/abi_smoke.c (plain C99, gcc) against include/sdfr.h and the in-tree libsdfr_hip.so; returns the binary path."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c")
    exe = os.path.join(str(out_dir), "abi_smoke")
    libdir = os.path.join(ROOT, "sdflabel_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wno-unused-result", src, "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", "-L" + libdir, "-lsdfr_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def pattern_weights(shape, salt):
    """Deterministic pseudo-random weights in [-1, 1] (integer hash of the flat index: exact in float32 on every platform); the same
    function as tools/make_golden.py's, so that the gradient functionals of the large goldens (G10) need no stored weight arrays."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    h = (h >> np.uint64(7)) % np.uint64(2001)
    return ((h.astype(np.int64) - 1000).astype(np.float32) / np.float32(1000.0)).reshape(shape)
