#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sphere_tracer.py tests/test_gpu_traced_refine.py tests/test_gpu_memory.py -q 2>&1 | tail -40 > $O/pytest_b.log
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --kw '[{}, {"uniform_tiles": false}, {"tail_rows": 0, "uniform_tiles": false}, {"head_steps": 4}, {"head_steps": 6}, {"cone_steps": 14}, {"cone_block": 8}]' > $O/sphere_b.log 2>&1
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --size 512 --steps 256 --kw '[{}, {"uniform_tiles": false}]' >> $O/sphere_b.log 2>&1
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --batch 8 --kw '[{}, {"uniform_tiles": false}]' >> $O/sphere_b.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b -o t -- python $R/tools/sphere_time.py --only f16 --cone 4 --kw '[{}]' --reps 3 > $O/trace_b.log 2>&1
cd $R
python tools/trace_gaps.py $O/trace_b sdfr_trace_cone_setup_kernel > $O/gaps_b.txt 2>&1
rm -rf $O/trace_b
tail -30 $O/pytest_b.log; cat $O/sphere_b.log | grep float16; cat $O/gaps_b.txt | tail -90
