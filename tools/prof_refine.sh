#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refine_$TAG -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_refine_$TAG.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_refine_$TAG/trace_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
