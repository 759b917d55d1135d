#!/bin/bash
# Median kernel times (rocprofv3 kernel trace) of one-stream launch groups + the bench value, for the library in place.
# GPU box:  bash tools/lds_tuning/kernel_times.sh   -> gpurun_out/dedupx/summary.txt   (EXTRA='--batch 8000 ...' for other shapes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
OUT=$R/gpurun_out/dedupx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for X in 0; do
  rm -rf /tmp/px_$X
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$X -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-weave --no-verify --steps 8 --warmup 2 --min-seconds 0.3 $EXTRA > $OUT/x$X.json 2> $OUT/x$X.err < /dev/null
  echo "== X=$X" >> $OUT/summary.txt
  python3 - /tmp/px_$X >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'lg::' not in n: continue
    key = (n[:n.rfind('(')].replace('float __vector(4)', 'float4').replace(' ', '')[:56], r['Grid_Size_X'], r['Grid_Size_Y'])
    acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:22]:
    v = sorted(v)
    print(k, len(v), "median %.1f" % v[len(v)//2])
PY
  python3 -c "import json,sys; d=json.loads(open('$OUT/x$X.json').read().strip().splitlines()[-1]); print('value', d['value']/1e9, 'ms', d['ms_per_step'])" >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
