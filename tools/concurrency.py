"""Kernel concurrency histogram of the hipGraph-timed region of a bench.py rocprofv3 trace."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
presc, steps, warm = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tr = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('lg::') or 'gather_kernel' in r['Kernel_Name']]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:30], r['Queue_Id']) for r in tr)
bg = [k for k in ks if 'batch_generate' in k[2]]
lo = presc + steps + warm + steps // 5
hi = presc + steps + warm + steps - steps // 5
t0, t1 = bg[lo][0], bg[hi][0]
win = [k for k in ks if t0 <= k[0] < t1]
ev = sorted([(k[0], 1) for k in win] + [(k[1], -1) for k in win])
cur, last, conc = 0, None, collections.Counter()
for t, d in ev:
    if last is not None:
        conc[cur] += t - last
    cur += d
    last = t
tot = sum(k[1] - k[0] for k in win)
n = hi - lo
print(f"{(t1 - t0) / n / 1e3:.1f} us/batch wall; {tot / n / 1e3:.1f} us/batch summed kernel time; queues {sorted(set(k[3] for k in win))}")
print("time share by #concurrent kernels:", {k: f"{v / (t1 - t0) * 100:.0f}%" for k, v in sorted(conc.items())})
