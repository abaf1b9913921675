"""From a `rocprofv3 --kernel-trace` dump (…_kernel_trace.csv): for one steady-state training step (Adam launch to Adam
launch), how long the chip ran with how many wavefronts in flight, and which kernels were running in the thin periods
(< 1024 wavefronts = less than one per SIMD): where a latency-bound chain leaves the chip idle."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
step_no = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = []
for r in rows:
    wg = 1
    for ax in "XYZ":
        wg *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
    threads = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:44]
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), wg * ((threads + 63) // 64), nm))
K.sort()
ends = [k[1] for k in K if k[3].startswith("adam_kernel")]
t0, t1 = ends[step_no], ends[step_no + 1]
step = [k for k in K if k[0] >= t0 and k[0] < t1]
print(f"step {step_no}: {(t1 - t0) / 1e3:.1f} us wall, {len(step)} launches, {sum(k[1] - k[0] for k in step) / 1e3:.1f} us of kernel time")
ev = []
for s, e, w, nm in step:
    ev += [(s, 1, w, nm), (e, -1, w, nm)]
ev.sort()
waves, active, last, hist, thin = 0, {}, t0, {}, {}
for t, d, w, nm in ev:
    dt = t - last
    if dt > 0:
        b = "idle" if waves == 0 else "< 1024 wavefronts" if waves < 1024 else "< 4096" if waves < 4096 else ">= 4096"
        hist[b] = hist.get(b, 0) + dt
        if waves < 1024:
            key = tuple(sorted(active)) or ("(nothing)",)
            thin[key] = thin.get(key, 0) + dt
    waves += d * w
    active[nm] = active.get(nm, 0) + d
    if active[nm] == 0:
        del active[nm]
    last = t
print({k: round(v / 1e3, 1) for k, v in hist.items()})
for key, dt in sorted(thin.items(), key=lambda x: -x[1])[:22]:
    print(f"{dt / 1e3:7.1f} us  {', '.join(key)}")
