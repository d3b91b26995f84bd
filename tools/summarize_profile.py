"""Condense a gpurun_out rocprofv3 session (tools/gpu_round.sh) into the tracked summaries under profiles/.

usage: python tools/summarize_profile.py <tag> [<round-name>]
  gpurun_out/prof_<tag>/trace_kernel_stats.csv            -> profiles/<round>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
  gpurun_out/pmc_{fetch,write}_<tag>/pmc_counter_collection.csv -> profiles/<round>_pmc_hbm.json (separate --pmc passes)
  and profiles/traffic_mlp_forward.json (read by bench.py for roofline.traffic)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) is doubled on gfx950 for wide coalesced reads,
WRITE_SIZE (KB) is taken as is (it reproduces the decoder's 64000 x 4 B output exactly).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else tag
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

src = os.path.join(G, "prof_%s" % tag, "trace_kernel_stats.csv")
rows = list(csv.DictReader(open(src)))
with open(os.path.join(P, "%s_kernel_stats.csv" % rnd), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"].split("(")[0][:100]
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

pmc = {}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(G, "pmc_%s_%s" % (kind, tag), "pmc_counter_collection.csv")
    if not os.path.isfile(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "sdfr" in r["Kernel_Name"][:12] and r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[cname + "_KB_mean"] = sum(v) / len(v)
        pmc[k]["launches_" + kind] = len(v)
for k, d in pmc.items():
    d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE_KB_mean", 0.0) + d.get("WRITE_SIZE_KB_mean", 0.0)) * 1024.0
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 5 --warmup 2 "
                   "--no-cpu-baseline`; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction)", "kernels": pmc},
          open(os.path.join(P, "%s_pmc_hbm.json" % rnd), "w"), indent=1)
key = sorted([k for k in pmc if "sdfr_mlp_kernel" in k], key=lambda k: -pmc[k].get("FETCH_SIZE_KB_mean", 0.0))[:1]   # the grid forward
if key:
    json.dump({"kernel": key[0], "source": "%s_pmc_hbm.json" % rnd, "hbm_bytes_per_launch": pmc[key[0]]["hbm_bytes_per_launch"]},
              open(os.path.join(P, "traffic_mlp_forward.json"), "w"), indent=1)
for name in ("bench_%s.json" % tag,):
    if os.path.isfile(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, "%s_bench.json" % rnd))
print("wrote profiles/%s_*" % rnd)
