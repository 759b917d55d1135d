"""Folds the kernel trace of a `sampling_server` run in the `slab` hand-over (tools/server_throughput.py --profile-server DIR --modes slab)
into: per-batch hand-over gather launches (duration, start-to-start interval, idle gap between consecutive ones), and how much of the
time other kernels (the groups' sampler phases) run beside them.   python tools/slab_trace.py <dir>/slab_b8000"""
import csv, glob, statistics, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_Y", 1) or 1)) for r in csv.DictReader(open(f)) if "lg::" in r["Kernel_Name"]]
rows.sort()
ho = [r for r in rows if "gather_kernel" in r[2] and r[3] == 1 and r[2][:r[2].rfind("(")].rstrip().endswith("true>")]
ho = ho[len(ho) // 4:]                     # steady state
others = [r for r in rows if r[3] > 1]
dur = [(e - s) / 1e3 for s, e, _, _ in ho]
s2s = [(b[0] - a[0]) / 1e3 for a, b in zip(ho, ho[1:])]
gap = [max(0, b[0] - a[1]) / 1e3 for a, b in zip(ho, ho[1:])]
t0, t1 = ho[0][0], ho[-1][1]
busy_others = sum(min(e, t1) - max(s, t0) for s, e, _, _ in others if e > t0 and s < t1) / 1e3
print(f"{len(ho)} hand-over gathers: duration median {statistics.median(dur):.1f} us (p10 {sorted(dur)[len(dur)//10]:.1f}, p90 {sorted(dur)[len(dur)*9//10]:.1f}); "
      f"start-to-start median {statistics.median(s2s):.1f} us = {1e6 / statistics.median(s2s):.0f} batches/s; idle between consecutive gathers median {statistics.median(gap):.1f} us, mean {sum(gap)/len(gap):.1f} us")
print(f"group kernels (sampler phases) busy {busy_others:.0f} us of the {(t1 - t0) / 1e3:.0f} us window = {busy_others / ((t1 - t0) / 1e3):.2f}; by kernel:")
acc = {}
for s, e, n, gy in others:
    if e > t0 and s < t1:
        k = n[:n.rfind("(")].replace("void ", "").replace("lg::", "")[:40]
        acc[k] = acc.get(k, 0) + (e - s) / 1e3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
    print(f"   {k}: {v:.0f} us")
