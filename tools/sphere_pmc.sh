#!/bin/bash
# PMC passes over the sphere-tracing mode (f16 decoder, default schedule): HBM bytes per launch (FETCH_SIZE, WRITE_SIZE in separate passes) and the
# matrix-pipe utilisation of the march kernels (SQ counters, a third pass).  No trace domains combined with --pmc.  tools/sphere_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-r03}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmcsph_${name}_$TAG -o pmc -- python $R/tools/sphere_time.py --only f16 --spec 4 > $O/pmcsph_${name}_$TAG.log 2>&1
  grep -i "error code\|exceeds" $O/pmcsph_${name}_$TAG.log | head -2
done
python - <<PY
import csv, collections, json
out = collections.defaultdict(dict)
for name in ("fetch", "write", "sq"):
    try:
        rows = list(csv.DictReader(open("$O/pmcsph_%s_$TAG/pmc_counter_collection.csv" % name)))
    except Exception as e:
        print(name, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "sdfr" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][:90]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_us_" + name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in agg.items():
        for c, v in d.items():
            # launches that exit at once (device-side gates) would dilute the means: keep the launches that ran (> 10 us)
            dur = d["duration_us_" + name]
            keep = [x for x, t in zip(v, dur) if t > 10.0] or v
            out[k][c + "_mean"] = sum(keep) / len(keep)
            out[k]["launches_that_ran_" + name] = len(keep)
for k, d in out.items():
    if "FETCH_SIZE_mean" in d or "WRITE_SIZE_mean" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE_mean", 0.0) + d.get("WRITE_SIZE_mean", 0.0)) * 1024.0
        d["hbm_GBps"] = d["hbm_bytes_per_launch"] / (d.get("duration_us_fetch_mean", 1.0) * 1e-6) / 1e9
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in d and d.get("SQ_BUSY_CYCLES_mean"):
        d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / d["SQ_BUSY_CYCLES_mean"]
json.dump({"note": "rocprofv3 --pmc passes over tools/sphere_time.py --only f16 --spec 4 (one 256x256 crop, float16 decoder, default schedule); means over the "
           "launches that actually ran (> 10 us; the march gates its launches on the device); hbm bytes = (2 FETCH_SIZE + WRITE_SIZE) KB", "kernels": out},
          open("$O/pmc_sphere_$TAG.json", "w"), indent=1)
for k, d in out.items():
    print(k[:80], {c: round(v, 2) for c, v in d.items() if c in ("hbm_GBps", "mfma_busy_over_sq_busy", "duration_us_fetch_mean", "hbm_bytes_per_launch", "launches_that_ran_fetch")})
PY
