#!/bin/bash
# one rocprofv3 --pmc pass over an arbitrary python script, per-kernel means: tools/pmc_any.sh <tag> <script.py> COUNTER...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=$1; SCRIPT=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 90 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$TAG -o pmc -- python $R/$SCRIPT > $O/pmc_$TAG.log 2>&1
grep -i "error code\|exceeds" $O/pmc_$TAG.log | head -2
python - <<PY
import csv, collections
try:
    rows = list(csv.DictReader(open("$O/pmc_$TAG/pmc_counter_collection.csv")))
except Exception as e:
    print("$TAG", e); rows = []
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "sdfr_" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   ", c, "mean=%.4g n=%d" % (sum(v) / len(v), len(v)))
PY
