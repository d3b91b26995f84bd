#!/bin/bash
# rocprofv3 kernel stats of a short bench run, top kernels only: tools/kstats.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-k}; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $O/ks_$TAG.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/ks_$TAG/**/ks_kernel_stats.csv", recursive=True) + glob.glob("$O/ks_$TAG/ks_kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-70s %5s %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
