#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_$TAG.log
timeout 600 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_write_$TAG.log 2>&1
cd $R
cat $O/pytest_$TAG.log; cat $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
find $O/prof_$TAG -name "*stats*" | head; 
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
