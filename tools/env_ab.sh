#!/bin/bash
# Whole-job A/B of ONE environment variable on one box, alternating runs (no profiler):
#   tools/env_ab.sh LEGION_WEAVE_PRIORITY -1 0 [rounds=3] [extra bench.py args]   -> gpurun_out/env_ab_<var>.txt
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
OUT=$R/gpurun_out/env_ab_$VAR.txt; mkdir -p $R/gpurun_out; : > $OUT
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v timeout -k 5 400 python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 "$@" 2> /tmp/env_ab.err < /dev/null | tail -1 > /tmp/env_ab.json
    python3 - $VAR $v >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/env_ab.json').read())
    print(f"{sys.argv[1]}={sys.argv[2]}: {d['value'] / 1e9:.3f} G edges/s, {d['ms_per_step']:.4f} ms/step, gather frac {d['roofline']['frac']:.3f}, sampler-only {d['sampling_only']['edges_per_sec'] / 1e9:.2f} G")
except Exception as e:
    print(f"{sys.argv[1]}={sys.argv[2]}: FAILED {e}", open('/tmp/env_ab.err').read()[-400:])
PY
    tail -1 $OUT
  done
done
