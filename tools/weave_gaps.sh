#!/bin/bash
# Under the weave default: how long the heavy stream idles between the last gather of group k and the first kernel of the rest of
# group k+1 (= the head of k+1 was not done in time), and how busy the heavy stream is overall.   bash tools/weave_gaps.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/px_w
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/px_w -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify --steps 8 --warmup 2 --min-seconds 0.3 $EXTRA > /dev/null 2> /dev/null < /dev/null
python3 - /tmp/px_w <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'lg::' not in n or not r['Grid_Size_Y'].isdigit() or int(r['Grid_Size_Y']) < 32: continue
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n[:n.rfind('(')].replace('float __vector(4)', 'float4').replace(' ', ''), r['Queue_Id'] if 'Queue_Id' in r else r.get('Stream_Id', '')))
rows.sort()
gaps = []
lasts = [(s, e) for s, e, n, q in rows if 'gather_kernel' in n and n.endswith('true>')]
lasts = lasts[len(lasts) // 2:]                       # the timed regions (replays), not the counting pass
for (s0, e0), (s1, e1) in zip(lasts, lasts[1:]):
    inside = [(s, e, n) for s, e, n, q in rows if e0 <= s < s1]
    first_rest = min((s for s, e, n in inside if 'sample_kernel' in n or 'dedup' in n or 'compact' in n or 'gather' in n), default=s1)
    gaps.append(((first_rest - e0) / 1e3, (s1 - e0) / 1e3, (e1 - s1) / 1e3))
gaps.sort()
g = gaps[len(gaps) // 2]
print("median over %d groups: idle after the last gather %.1f us; last gather(k) end -> last gather(k+1) start %.1f us; last gather %.1f us" % (len(gaps), g[0], g[1], g[2]))
PY
