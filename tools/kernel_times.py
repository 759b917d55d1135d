"""Prints avg duration per kernel name (and the hop-2 sample kernel separately) from a rocprofv3 kernel trace."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if n.startswith('lg::') or 'gather_kernel' in n:
        d[n.split('(')[0][:40]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 50: continue
    v2 = sorted(v)
    print(f"{k:42s} n={len(v):5d} avg={sum(v)/len(v)/1e3:7.1f}us p50={v2[len(v)//2]/1e3:7.1f} p90={v2[len(v)*9//10]/1e3:7.1f} max={v2[-1]/1e3:7.1f}")
