"""Shared helpers for the tests (fixtures, decoder loading).  Never reads /root/reference."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ASSET = os.path.join(ROOT, "sdflabel_amd", "assets", "deepsdf_synth")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def fitted_state():
    """The committed synthetic 8x512 decoder as {key: float32 ndarray} + its NetworkSpecs."""
    import torch
    st = torch.load(ASSET + ".pt", map_location="cpu")["model_state_dict"]
    st = {k[len("module."):] if k.startswith("module.") else k: v.float().numpy() for k, v in st.items()}
    spec = json.load(open(ASSET + ".json"))["NetworkSpecs"]
    return st, spec


def K_for(H, W):
    f = 45.0 * H / 32.0
    return np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)


def state_from_npz(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}
