"""Merge the rocprofv3 --pmc passes of tools/pmc_any.sh (gpurun_out/pmc_<prefix><x>_<tag>/pmc_counter_collection.csv, one counter group per pass)
into one JSON of per-kernel means:  python tools/summarize_pmc.py <prefix> <tag> <out.json> [kernel-name substring]
e.g. python tools/summarize_pmc.py jac16 r06 profiles/r06_pmc_jac16.json Li3E     (the MODE-3 = mask-fed Jacobian instantiations)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
want = sys.argv[4] if len(sys.argv) > 4 else "sdfr_"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
passes = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s*_%s" % (prefix, tag), "pmc_counter_collection.csv")))
for path in passes:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "sdfr_" not in name or want not in name:
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[name]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        agg[name]["_meta"] = [{"workgroup_size": int(r["Workgroup_Size"]), "grid_size": int(r["Grid_Size"]), "lds_bytes": int(r["LDS_Block_Size"]),
                               "vgprs": int(r["VGPR_Count"]), "agprs": int(r["Accum_VGPR_Count"]), "sgprs": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"])}]
res = {}
for k, d in agg.items():
    meta = d.pop("_meta")[0]
    res[k] = dict(meta, launches_per_pass=len(d["duration_us"]) // max(1, len(passes)), **{c: sum(v) / len(v) for c, v in sorted(d.items())})
    g = res[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (profiles/r04_pmc_f16_forward.json: the same
        # normalisation reproduces busy = 32 cycles x MFMA instructions there)
        g["cycles_per_xcd"] = g["GRBM_GUI_ACTIVE"] / 8.0
        g["clock_GHz_under_profiler"] = g["cycles_per_xcd"] / (g["duration_us"] * 1e3)
        g["mfma_busy_frac"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["cycles_per_xcd"] * 1024.0)
    if "SQ_INSTS_VALU" in g and "SQ_INSTS_MFMA" in g and g["SQ_INSTS_MFMA"]:
        g["valu_per_mfma"] = g["SQ_INSTS_VALU"] / g["SQ_INSTS_MFMA"]
    if "TCC_HIT_sum" in g and "TCC_REQ_sum" in g and g["TCC_REQ_sum"]:
        g["l2_hit_rate"] = g["TCC_HIT_sum"] / g["TCC_REQ_sum"]
    if "SQ_LDS_BANK_CONFLICT" in g and "SQ_LDS_IDX_ACTIVE" in g and g["SQ_LDS_IDX_ACTIVE"]:
        g["lds_bank_conflict_frac"] = g["SQ_LDS_BANK_CONFLICT"] / g["SQ_LDS_IDX_ACTIVE"]
json.dump({"note": "rocprofv3 --pmc, one counter group per pass (tools/pmc_any.sh); per-kernel means over the launches of each pass",
           "passes": [os.path.relpath(p, ROOT) for p in passes], "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, list(res))
