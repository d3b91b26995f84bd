#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > $O/pytest_e.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/pytest_e.log | cut -c1-500 | head -80
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --kw '[{}, {"cone_spec_k": 1, "cone_steps": 10}, {"cone_spec_k": 2}, {"cone_spec_k": 4, "cone_steps": 3}, {"cone_spec_k": 4, "cone_steps": 5}, {"spec_from": 4, "spec_from2": 7}, {"spec_from": 6, "spec_from2": 9}, {"spec_from": 8, "spec_from2": 11}, {"spec_from": 6, "spec_from2": 8}, {"spec_from": 5, "spec_from2": 9}, {"spec_from": 3, "spec_from2": 6}]' > $O/sphere_e.log 2>&1
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --size 512 --steps 256 --kw '[{}, {"cone_spec_k": 1, "cone_steps": 10}, {"spec_from": 8, "spec_from2": 11}, {"spec_from": 10, "spec_from2": 13}]' >> $O/sphere_e.log 2>&1
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --batch 8 --kw '[{}, {"cone_spec_k": 1, "cone_steps": 10}, {"spec_from": 6, "spec_from2": 9}]' >> $O/sphere_e.log 2>&1
grep float16 $O/sphere_e.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_e -o t -- python $R/tools/sphere_time.py --only f16 --cone 4 --kw '[{}]' --reps 3 > $O/trace_e.log 2>&1
cd $R
python tools/trace_gaps.py $O/trace_e sdfr_trace_cone_setup_kernel > $O/gaps_e.txt 2>&1
rm -rf $O/trace_e
cat $O/gaps_e.txt | awk '{print $1, $3, $5}' | cut -c1-70 | tail -60
