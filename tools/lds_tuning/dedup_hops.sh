#!/bin/bash
# duration of every kernel of a launch group by launch order (hop 1 and hop 2 share a grid for some of them); one stream (--no-weave)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/px_h
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/px_h -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-weave --no-verify --steps 8 --warmup 2 --min-seconds 0.3 $EXTRA > /dev/null 2> /dev/null < /dev/null
python3 - /tmp/px_h <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted((int(r['Start_Timestamp']), r['Kernel_Name'][:r['Kernel_Name'].rfind('(')].replace('float __vector(4)', 'float4').replace(' ', '')[-44:], int(r['Grid_Size_Y']) if r['Grid_Size_Y'].isdigit() else 0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if 'lg::' in r['Kernel_Name'])
rows = [r for r in rows if r[2] >= 32]
# the last complete group: from the last batch_generate onwards
starts = [i for i, r in enumerate(rows) if 'batch_generate' in r[1]]
seq = collections.defaultdict(list)
for a, b in zip(starts[-40:-1], starts[-39:]):
    for j, r in enumerate(rows[a:b]):
        seq[(j, r[1])].append(r[3])
tot = 0
for (j, name), v in sorted(seq.items()):
    d = sorted(v)
    print(f"{j:2d} {name:44s} n={len(v):3d} median {d[len(d)//2]:8.1f} us")
    tot += d[len(d)//2]
print("sum of medians %.1f us" % tot)
PY
