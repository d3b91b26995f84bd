from sdflabel_amd.deepsdf.workspace import setup_dsdf  # noqa: F401
