#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ragged.py -q 2>&1 > $O/pytest_d_ragged.log
grep -n "^E  \|^FAILED\|passed\|failed\|Error" $O/pytest_d_ragged.log | cut -c1-600 | head -60
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ragged.py 2>&1 > $O/pytest_d_all.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/pytest_d_all.log | cut -c1-400 | head -60
timeout 900 python bench.py --total-crops 128 --configs4-crops 64 > $O/bench_d.json 2> $O/bench_d.err
tail -3 $O/bench_d.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/bench_d.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"])
print(json.dumps(b["optimizer_mirror_varied_crops"], indent=1))
print(json.dumps(b["refine_demo_traced"], indent=1))
for k, v in b["sphere_trace"].items():
    print(k, v.get("ms_per_render_fwd_bwd"), v.get("march_ms"), v.get("roofline_march", {}).get("frac"), v.get("error"))
PY
