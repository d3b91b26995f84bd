#!/bin/bash
# The splat pair (and the other per-crop kernels) at 64 crops per launch: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ counters
# that say which unit binds.  --pmc passes only, never combined with trace domains.  Summary -> gpurun_out/pmc_splat64_<tag>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
pass() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/s64_${name}_$TAG -o pmc -- $CMD > $O/s64_${name}_$TAG.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python - <<PY
import csv, collections, json, os
out = {}
for name in ("fetch", "write", "sq1", "sq2"):
    f = "$O/s64_%s_$TAG/pmc_counter_collection.csv" % name
    if not os.path.isfile(f):
        out.setdefault("_missing", []).append(name); continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("splat", "loss", "surfels", "pose_latent", "band")):
            continue
        d = out.setdefault(k[-70:], {})
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        d.setdefault("_dur_us_" + name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if c in r: d[c] = r[c]
res = {k: {c: (sum(v) / len(v) if isinstance(v, list) else v) for c, v in d.items()} for k, d in out.items() if isinstance(d, dict)}
res["_note"] = "means per launch over the profiled steps of: " + "$CMD".replace("$R/", "") + " (64 crops per launch); SQ counters in quad-cycles summed over all waves / SEs as rocprofv3 reports them; FETCH_SIZE / WRITE_SIZE in the counter's units (see the guide's corrections in tools/summarize_profile.py)"
json.dump(res, open("$O/pmc_splat64_$TAG.json", "w"), indent=1)
for k, d in res.items():
    if isinstance(d, dict): print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items()})
PY
find $O -name "*_kernel_trace.csv" -path "*s64_*" -delete; find $O -name "*.db" -path "*s64_*" -delete
