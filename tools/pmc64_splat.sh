R=${GRAFT_REPO_ROOT}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/x_fetch64 -o pmc -- python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/x_fetch64.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/x_write64 -o pmc -- python $R/bench.py --crops-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/x_write64.log 2>&1
python - <<PY
import csv, collections
for name, c in (("fetch64","FETCH_SIZE"),("write64","WRITE_SIZE")):
    rows=list(csv.DictReader(open("$O/x_%s/pmc_counter_collection.csv"%name)))
    agg=collections.defaultdict(list); dur=collections.defaultdict(list)
    for r in rows:
        k=r["Kernel_Name"].split("(")[0][:50]
        if "splat" in k or "loss" in k or "project" in k or "band" in k or "pose_latent" in k:
            agg[k].append(float(r["Counter_Value"])); dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in agg.items(): print(name, k, "KB mean %.0f"%(sum(v)/len(v)), "us %.1f"%(sum(dur[k])/len(dur[k])), "n", len(v))
PY
