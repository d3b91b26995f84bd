from sdflabel_amd.renderer.utils_rasterer import qrot, calibration_matrix  # noqa: F401
