#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sphere_tracer.py tests/test_gpu_traced_refine.py -q 2>&1 > $O/pytest_c_full.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/pytest_c_full.log | cut -c1-400 | head -120
timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_sphere_tracer.py --deselect tests/test_gpu_traced_refine.py 2>&1 | tail -8
