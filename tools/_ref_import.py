"""Import helper for the upstream reference (THIS container only).

The reference tree at /root/reference is read-only and never travels to the GPU
box.  This module is used only by the golden-vector generators under tools/ --
never by the product package, tests, bench.py or smoke().

It (a) disables bytecode writing so nothing is written into the reference tree,
(b) stubs the GUI / IO modules the reference imports at module scope but never
touches on the hot path (open3d, cv2, pyquaternion), (c) puts the reference's
two import roots on sys.path (README.md:21-24 of the reference).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

REF_ROOT = "/root/reference"


def setup():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden generation only runs in the build container")
    for name in ("open3d", "cv2", "pyquaternion"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    for p in (os.path.join(REF_ROOT, "sdfrenderer"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
