#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > $O/pytest_f.log
grep -n "^FAILED\|passed\|failed" $O/pytest_f.log | cut -c1-300 | head -40
grep -n "^E  " $O/pytest_f.log | grep -v "+  " | cut -c1-300 | head -30
timeout 300 python tools/diag_cropped.py 2>&1 | tail -24
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --kw '[{}, {"spec_from": 5, "spec_from2": 8}, {"spec_from": 4, "spec_from2": 8}, {"spec_from": 6, "spec_from2": 10}, {"cone_spec_k": 8, "cone_steps": 3}, {"cone_block": 8}, {"uniform_tiles": false}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --size 512 --steps 256 --kw '[{}, {"spec_from": 7, "spec_from2": 11}, {"spec_from": 9, "spec_from2": 13}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --size 128 --kw '[{}, {"spec_from": 4, "spec_from2": 8}, {"spec_from": 5, "spec_from2": 9}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --batch 8 --kw '[{}, {"spec_from": 8, "spec_from2": 12}]' 2>&1 | grep float16
timeout 300 python tools/sphere_time.py --only f16 --cone 4 --batch 64 --reps 3 --kw '[{}, {"spec_from": 10, "spec_from2": 13}, {"cone_spec_k": 1, "cone_steps": 10}]' 2>&1 | grep float16
timeout 900 python bench.py --total-crops 128 --configs4-crops 64 > $O/bench_f.json 2> $O/bench_f.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/bench_f.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"])
d = b["dropin_api"]; print("dropin", d["ms_per_step"], d["launches"].get("host_syncs"), d["launches"].get("library_hip_kernels"), d["launches"].get("library_torch_glue"))
print("traced", b["refine_demo_traced"]["value"], b["refine_demo_traced"]["ms_per_iteration"], b["refine_demo_traced"]["yaw_error_before_after"])
print("prefilter", b["prefilter_decoder"].get("ms_per_step"), b["prefilter_decoder"].get("ms_per_step_without_audit"), b["prefilter_decoder"].get("guard"))
for k, v in b["sphere_trace"].items():
    print(k, v.get("ms_per_render_fwd_bwd"), v.get("march_ms"), v.get("roofline_march", {}).get("frac"), v.get("ray_evaluations"), v.get("error"))
PY
