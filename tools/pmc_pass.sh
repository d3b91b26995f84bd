#!/bin/bash
# one rocprofv3 --pmc pass over the decoder forward alone: tools/pmc_pass.sh <tag> <f32|f16|split> COUNTER...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=$1; PREC=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$TAG -o pmc -- python $R/tools/mlp_only.py 6 $PREC > $O/pmc_$TAG.log 2>&1
grep -i "error\|invalid\|fail" $O/pmc_$TAG.log | head -3
python - <<PY
import csv, collections
try:
    rows = list(csv.DictReader(open("$O/pmc_$TAG/pmc_counter_collection.csv")))
except Exception as e:
    print("$TAG", e); rows = []
agg = collections.defaultdict(list)
for r in rows:
    if "sdfr_mlp" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print("$TAG", k, "mean=%.4g" % (sum(v[1:]) / max(len(v) - 1, 1)))
PY
