#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > $O/pytest_g.log
grep -n "^FAILED\|passed\|failed" $O/pytest_g.log | cut -c1-300 | head -40
grep -n "^E  " $O/pytest_g.log | grep -v "+  " | cut -c1-400 | head -30
timeout 300 python tools/sphere_time.py --only f16 --kw '[{}, {"spec_from": 10, "spec_from2": 13}, {"spec_from": 7, "spec_from2": 11}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --size 512 --steps 256 --kw '[{}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --size 128 --kw '[{}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --batch 64 --reps 3 --kw '[{}]' 2>&1 | grep float16
