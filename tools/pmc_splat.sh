#!/bin/bash
# HBM bytes and duration of the splat kernels at 64 crops per launch (BASELINE configs[2] shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_splat_${c}_$TAG -o pmc -- python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_splat_${c}_$TAG.log 2>&1
done
python - <<PY
import csv, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows = list(csv.DictReader(open("$O/pmc_splat_%s_$TAG/pmc_counter_collection.csv" % c)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "sdfr_splat" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            agg[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"])))
    for k, v in agg.items():
        print(c, k, "n=%d KB=%.0f dur_us=%.1f grid=%d" % (len(v), sum(x[0] for x in v)/len(v), sum(x[1] for x in v)/len(v)/1e3, v[0][2]))
PY
