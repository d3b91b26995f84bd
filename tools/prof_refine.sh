#!/bin/bash
# kernel time table of one sharded refinement (tools/refine_sharded.py): tools/prof_refine.sh <tag> [refine_sharded.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refine_$TAG -o trace -- python $R/tools/refine_sharded.py "$@" > $O/prof_refine_$TAG.log 2>&1
tail -2 $O/prof_refine_$TAG.log
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_refine_$TAG/trace_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $O/prof_refine_$TAG -name "*_kernel_trace.csv" -delete; find $O/prof_refine_$TAG -name "*.db" -delete
