import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (it is git-ignored): compile it once with hipcc (cross-compiles without a GPU)
    lib = os.path.join(ROOT, "sdflabel_amd", "lib", "libsdfr_hip.so")
    if not os.path.isfile(lib) and os.path.isfile("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "sdflabel_amd", "csrc", "build.sh")])


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
