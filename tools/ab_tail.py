"""A/B the single-variant libraries sdflabel_amd/lib/ab/libsdfr_t*.so (tail-kernel geometries, tools/ab_variant.sh) on the f16 sphere march"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.join(ROOT, "sdflabel_amd", "lib", "libsdfr_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "sdflabel_amd", "lib", "ab", "libsdfr_t*.so")))
for lib in libs:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sphere_time.py"), "--only", "f16", "--spec", "4"], env=dict(os.environ, SDFR_LIB=lib),
                         capture_output=True, text=True)
    print(os.path.basename(lib), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
