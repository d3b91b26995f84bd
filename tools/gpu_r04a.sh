#!/bin/bash
# r04 session A: the whole GPU suite (new traced-refinement tests included), tracer timings, default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_traced_refine.py 2>&1 | tail -15 > $O/pytest_a_old.log
timeout 900 python -m pytest tests/test_gpu_traced_refine.py -q 2>&1 | tail -60 > $O/pytest_a_new.log
timeout 300 python tools/sphere_time.py --only f16 --cone 0 4 > $O/sphere_a.log 2>&1
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err
tail -5 $O/pytest_a_old.log; tail -40 $O/pytest_a_new.log; cat $O/sphere_a.log | tail -8; tail -c 1500 $O/bench_a.json; tail -3 $O/bench_a.err
