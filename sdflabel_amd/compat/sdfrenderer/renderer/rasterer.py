from sdflabel_amd.renderer.rasterer import Rasterer  # noqa: F401
