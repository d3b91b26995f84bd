"""GPU timeline of a kernel trace: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [anchor kernel substring] -- prints, for the
last repetition that starts at the anchor kernel, every kernel with its duration and the idle gap before it (development aid)."""
import csv, glob, os, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "sdfr_mlp_kernel<float, 32"
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if anchor in e[2]]
if len(starts) < 3:
    raise SystemExit("anchor not found often enough")
i0, i1 = starts[-3], starts[-2]
print("step of %d launches, %.1f us from anchor to anchor" % (i1 - i0, (ev[i1][0] - ev[i0][0]) / 1e3))
busy = gap_total = 0.0
prev_end = ev[i0 - 1][1] if i0 > 0 else ev[i0][0]
for s, e, n in ev[i0:i1]:
    gap = (s - prev_end) / 1e3
    print("%9.1f gap  %9.1f us  %s" % (gap, (e - s) / 1e3, n[:90]))
    busy += (e - s) / 1e3
    gap_total += max(gap, 0.0)
    prev_end = max(prev_end, e)
print("busy %.1f us, idle %.1f us" % (busy, gap_total))
