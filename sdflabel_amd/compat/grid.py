from sdflabel_amd.grid import Grid3D  # noqa: F401
