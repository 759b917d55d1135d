#!/bin/bash
# Whole-job A/B of library builds on one box, alternating:  ROUNDS=5 tools/lds_tuning/value_rounds.sh base x y   (variants v_<name>;
# base = the library in place).  Prints value (G edges/s) and ms per step per run, then the medians.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
V=$R/tools/lds_tuning/variants
OUT=$R/gpurun_out/value_rounds.txt; mkdir -p $R/gpurun_out; : > $OUT
for i in $(seq 1 ${ROUNDS:-5}); do
  for n in "$@"; do
    if [ $n = base ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$V/v_$n/liblegion_hip.so; fi
    unset LEGION_LDS_SMALL_BUCKETS LEGION_SAMPLE_MAX_WG LEGION_BENCH_ARENA LEGION_ARENA_ALIGN LEGION_ARENA_JITTER LEGION_SCATTER_LANES LEGION_SCATTER_ARENA LEGION_SCATTER_CHUNK_MB LEGION_ARENA_ORDER; [ -f $V/v_$n/env ] && . $V/v_$n/env
    echo -n "$n " >> $OUT
    timeout -k 5 200 python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify $EXTRA 2>/dev/null < /dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,4), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))" >> $OUT
  done
done
python3 - $OUT <<'PY'
import sys, collections, statistics
acc = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) >= 3: acc[f[0]].append((float(f[1]), float(f[2]), float(f[3])))
print(open(sys.argv[1]).read())
for k, v in acc.items():
    print(k, "median value %.4f G  ms/step %.4f  gather frac %.4f  (%d runs)" % (statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), statistics.median(x[2] for x in v), len(v)))
PY
