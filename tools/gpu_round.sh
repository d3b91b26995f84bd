#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r06}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_$TAG.log
timeout 900 python bench.py --extras $O/bench_extras_$TAG.json > $O/bench_$TAG.json 2> $O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats of the headline loop alone (--no-extras: every decoder-forward launch is the single-crop one the roofline is quoted on)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_$TAG.log 2>&1
# ... and of the whole default run (all informational sections; the sharded refinement shortened: its kernels are the refine_demo's at 64 crops)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proffull_$TAG -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --total-crops 128 --configs4-crops 64 > $O/proffull_$TAG.log 2>&1
# the sphere-tracing mode alone (march kernels: decoder forward on the active rows, step kernel, looping tail)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profsphere_$TAG -o trace -- python $R/tools/sphere_time.py --only f16 --spec 4 > $O/profsphere_$TAG.log 2>&1
# HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), headline loop only
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write_$TAG.log 2>&1
# the same two passes at 64 crops per launch (splat / Jacobian in the throughput regime)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch64_$TAG -o pmc -- python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch64_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write64_$TAG -o pmc -- python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write64_$TAG.log 2>&1
# the sphere march: HBM bytes and matrix-pipe counters (three --pmc passes)
bash $R/tools/sphere_pmc.sh $TAG > $O/pmcsph_$TAG.log 2>&1
# kernel time table of the float16 + candidate-reuse refinement (256 crops x 60 iterations in chunks of 64)
# (--serial-audit: per-kernel durations are only meaningful when the kernels do not overlap; the product default runs the audit chain beside the
# candidates' pass and the Jacobian, which stretches all three -- that run is the *_concurrent table)
bash $R/tools/prof_refine.sh ${TAG}_f16reuse --crops 256 --chunk 64 --precision float16 --reuse --serial-audit > $O/prof_refine_${TAG}_f16reuse.txt 2>&1
bash $R/tools/prof_refine.sh ${TAG}_f16reuse_concurrent --crops 256 --chunk 64 --precision float16 --reuse > $O/prof_refine_${TAG}_f16reuse_concurrent.txt 2>&1
bash $R/tools/prof_refine.sh ${TAG}_f32reuse --crops 128 --chunk 64 --precision float32 --reuse --serial-audit > $O/prof_refine_${TAG}_f32reuse.txt 2>&1
# the reference's shipped operating point (rendering_area 32, float16): per-annotation phases and a frame of 16 through optimize_many
bash $R/tools/prof_area32.sh ${TAG}_one --only phases > $O/prof_area32_${TAG}_one.txt 2>&1
bash $R/tools/prof_area32.sh ${TAG}_many16 --only many --frames 16 > $O/prof_area32_${TAG}_many16.txt 2>&1
timeout 300 python $R/tools/area32_time.py > $O/area32_$TAG.log 2>&1
# PMC passes over the float16 band Jacobian at 64 crops per launch (matrix-pipe busy, waits, L2, LDS)
bash $R/tools/pmc_jac16.sh $TAG > $O/pmc_jac16_$TAG.txt 2>&1
cd $R
# gpurun merges at most 64 MiB back: the raw per-launch traces (60 MB for the full run) are not read by tools/summarize_profile.py -- only the
# *_kernel_stats.csv of the trace runs and the *_counter_collection.csv of the PMC passes are
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
cat $O/pytest_$TAG.log | tail -3; tail -c 600 $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f"
