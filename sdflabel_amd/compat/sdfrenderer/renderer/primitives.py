from sdflabel_amd.renderer.primitives import inside_circle, inside_circle_opt, inside_surfel  # noqa: F401
