#!/bin/bash
# kernel time table of the per-annotation path at the reference's shipped operating point: tools/prof_area32.sh <tag> [area32_time.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_area32_$TAG -o trace -- python $R/tools/area32_time.py "$@" > $O/prof_area32_$TAG.log 2>&1
tail -3 $O/prof_area32_$TAG.log
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_area32_$TAG/trace_kernel_stats.csv")))
for r in rows[:40]:
    print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $O/prof_area32_$TAG -name "*_kernel_trace.csv" -delete; find $O/prof_area32_$TAG -name "*.db" -delete
