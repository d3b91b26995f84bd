#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; PREC=${2:-f32}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_$TAG -o pmc -- python $R/tools/mlp_only.py 6 $PREC > $O/pmc_sq_$TAG.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc_sq2_$TAG -o pmc -- python $R/tools/mlp_only.py 6 $PREC > $O/pmc_sq2_$TAG.log 2>&1
tail -3 $O/pmc_sq_$TAG.log; tail -3 $O/pmc_sq2_$TAG.log
python - <<PY
import csv, collections
for d in ("pmc_sq_$TAG","pmc_sq2_$TAG"):
    try:
        rows = list(csv.DictReader(open("$O/%s/pmc_counter_collection.csv" % d)))
    except Exception as e:
        print(d, e); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if "sdfr_mlp" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg["_dur_ns"].append(dur)
    for k, v in sorted(agg.items()):
        print(d, k, "mean=%.4g" % (sum(v[1:]) / max(len(v) - 1, 1)))
PY
