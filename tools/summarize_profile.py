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
sys.path.insert(0, ROOT)
from sdflabel_amd._lib import TRAFFIC_SOURCES, source_sha16  # noqa: E402  (kernel-source hash stored with every traffic figure: bench.py checks it)


def sha(name):
    return {"source_sha16": source_sha16(TRAFFIC_SOURCES[name]), "sources": list(TRAFFIC_SOURCES[name])}


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

srcf = os.path.join(G, "proffull_%s" % tag, "trace_kernel_stats.csv")
if os.path.isfile(srcf):
    with open(os.path.join(P, "%s_kernel_stats_all_sections.csv" % rnd), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in csv.DictReader(open(srcf)):
            w.writerow([r["Name"].split("(")[0][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

srcs = os.path.join(G, "profsphere_%s" % tag, "trace_kernel_stats.csv")
if os.path.isfile(srcs):
    with open(os.path.join(P, "%s_kernel_stats_sphere_f16.csv" % rnd), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in csv.DictReader(open(srcs)):
            w.writerow([r["Name"].split("(")[0][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

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
    json.dump(dict({"kernel": key[0], "source": "%s_pmc_hbm.json" % rnd, "hbm_bytes_per_launch": pmc[key[0]]["hbm_bytes_per_launch"]}, **sha("traffic_mlp_forward.json")),
              open(os.path.join(P, "traffic_mlp_forward.json"), "w"), indent=1)
# the same PMC pair at 64 crops per launch (tools/gpu_round.sh: pmc_fetch64_<tag> / pmc_write64_<tag>) and the splat pair's traffic
pmc64 = {}
for kind, cname in (("fetch64", "FETCH_SIZE"), ("write64", "WRITE_SIZE")):
    path = os.path.join(G, "pmc_%s_%s" % (kind, tag), "pmc_counter_collection.csv")
    if not os.path.isfile(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "sdfr" in r["Kernel_Name"][:12] and r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in agg.items():
        pmc64.setdefault(k, {})[cname + "_KB_mean"] = sum(x[0] for x in v) / len(v)
        pmc64[k]["duration_us_" + kind] = sum(x[1] for x in v) / len(v) / 1e3
for k, d in pmc64.items():
    d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE_KB_mean", 0.0) + d.get("WRITE_SIZE_KB_mean", 0.0)) * 1024.0
if pmc64:
    json.dump({"note": "as %s_pmc_hbm.json but over `python bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras` "
                       "(64 crops per launch)" % rnd, "kernels": pmc64}, open(os.path.join(P, "%s_pmc_hbm_64crops.json" % rnd), "w"), indent=1)


def splat_pair(d):
    f = [k for k in d if "sdfr_splat_fwd_kernel" in k]
    b = [k for k in d if "sdfr_splat_bwd_kernel" in k]
    return (d[f[0]]["hbm_bytes_per_launch"] + d[b[0]]["hbm_bytes_per_launch"]) if f and b else None


json.dump(dict({"source": "%s_pmc_hbm.json / %s_pmc_hbm_64crops.json" % (rnd, rnd), "what": "HBM bytes (2*FETCH_SIZE + WRITE_SIZE) of the splat forward + "
           "backward launch pair", "crops_1": splat_pair(pmc), "crops_64": splat_pair(pmc64) if pmc64 else None}, **sha("traffic_splat.json")),
          open(os.path.join(P, "traffic_splat.json"), "w"), indent=1)
for name in ("bench_%s.json" % tag,):
    if os.path.isfile(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, "%s_bench.json" % rnd))
print("wrote profiles/%s_*" % rnd)

sph = os.path.join(G, "pmc_sphere_%s.json" % tag)
if os.path.isfile(sph):
    shutil.copy(sph, os.path.join(P, "%s_pmc_sphere.json" % rnd))
    ks = json.load(open(sph))["kernels"]
    st = ks.get("sdfr_trace_step_kernel")
    if st and "hbm_bytes_per_launch" in st:
        # the march's advance / compaction kernel against the HBM roofline (read by bench.py: sphere_trace.*.step_kernel_hbm)
        json.dump({"source_sha16": sha("traffic_sphere_step.json")["source_sha16"], "source": "%s_pmc_sphere.json" % rnd, "what": "sdfr_trace_step_kernel, one 256x256 crop, float16 march: HBM bytes (2*FETCH_SIZE + WRITE_SIZE) "
                   "and duration per launch, means over the launches of the march (the active count shrinks from 65 k rays to a few thousand)",
                   "hbm_bytes_per_launch": st["hbm_bytes_per_launch"], "duration_us": st.get("duration_us_fetch_mean"), "GBps": st.get("hbm_GBps")},
                  open(os.path.join(P, "traffic_sphere_step.json"), "w"), indent=1)
