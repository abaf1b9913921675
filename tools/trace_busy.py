"""GPU busy / idle time from a rocprofv3 kernel trace: union of kernel intervals per step window.
    python tools/trace_busy.py <dir with *_kernel_trace.csv> [steps_in_trace]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
# steady-state window: from the first to the last adam kernel
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
n_use = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 1   # consecutive steps starting at the 2nd Adam launch
lo, hi = rows[adam[1]][1], rows[adam[1 + n_use - 1]][1]
steps = n_use - 2
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in sel)
print("steps %d  wall %.2f ms/step  union-busy %.2f ms/step (%.1f %%)  sum of kernels %.2f ms/step  kernels/step %d" % (
    steps + 1, (hi - lo) / 1e6 / (steps + 1), busy / 1e6 / (steps + 1), 100.0 * busy / (hi - lo), tot / 1e6 / (steps + 1), len(sel) // (steps + 1)))
# biggest idle gaps
gaps = []
cur_e = None
for s, e, n, _ in sel:
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, n))
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort(reverse=True)
print("idle total %.2f ms/step in %d gaps/step; largest gaps (us, kernel after the gap):" % (sum(g for g, _ in gaps) / 1e6 / (steps + 1), len(gaps) // (steps + 1)))
for g, n in gaps[:12]:
    print("  %7.1f  %s" % (g / 1e3, n[:90]))
# how much of the wall time has exactly one kernel in flight, and which kernels those are (the serial sections)
ev = []
for i, (s, e, n, _) in enumerate(sel):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
live, last, hist, alone = set(), None, {}, {}
for t, d, i in ev:
    if last is not None and t > last and live:
        k = min(len(live), 4)
        hist[k] = hist.get(k, 0) + (t - last)
        if len(live) == 1:
            nm = sel[next(iter(live))][2]
            nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            alone[nm] = alone.get(nm, 0) + (t - last)
    (live.add if d > 0 else live.discard)(i)
    last = t
print("time with k kernels in flight (ms/step): " + "  ".join("%s%d: %.2f" % ("" if k < 4 else ">=", k, v / 1e6 / (steps + 1)) for k, v in sorted(hist.items())))
print("alone on the chip (ms/step):")
for nm, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print("  %6.3f  %s" % (v / 1e6 / (steps + 1), nm))
