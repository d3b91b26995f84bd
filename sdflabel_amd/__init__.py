"""sdflabel_amd -- MI355X-native differentiable SDF renderer, drop-in for the renderer hot path of TRI-ML/sdflabel
(sdfrenderer.grid.Grid3D, sdfrenderer.renderer.rasterer.Rasterer, the DeepSDF decoder and setup_dsdf).

All arithmetic runs in hand-written HIP kernels behind the C ABI of include/sdfr.h (libsdfr_hip.so, bound with ctypes in
sdflabel_amd/_lib.py).  PyTorch supplies device memory, streams and autograd plumbing only.
"""
from ._lib import LIB_PATH, SdfrError, lib  # noqa: F401
from .grid import Grid3D  # noqa: F401
from .renderer.rasterer import Rasterer  # noqa: F401
from .deepsdf.workspace import setup_dsdf  # noqa: F401
from .deepsdf.networks.deep_sdf_decoder_scale import Decoder  # noqa: F401
from .batch import BatchRenderer  # noqa: F401
from .refine import BatchRefiner  # noqa: F401
from .renderer.sphere_tracer import SphereTracer  # noqa: F401

__all__ = ["Grid3D", "Rasterer", "Decoder", "setup_dsdf", "BatchRenderer", "BatchRefiner", "SphereTracer", "lib", "SdfrError", "LIB_PATH"]
