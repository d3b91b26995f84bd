import json, subprocess, sys, os
r = subprocess.run([sys.executable, "bench.py", "--crops-per-gpu", sys.argv[1], "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--extras", "/tmp/x.json"], capture_output=True, text=True)
e = json.load(open("/tmp/x.json")); s = e["roofline_splat"]
print(os.environ.get("SDFR_LIB", "product"), "crops", sys.argv[1], "fwd %.4f bwd %.4f ms" % (s["fwd_ms"], s["bwd_ms"]), "ms/step %.4f" % e["ms_per_step"])
