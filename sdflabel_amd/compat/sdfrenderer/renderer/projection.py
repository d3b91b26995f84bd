from sdflabel_amd.renderer.projection import project_in_2D, project_in_2D_quat  # noqa: F401
