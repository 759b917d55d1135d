#!/bin/bash
# Per-kernel timeline of a launch group (tools/trace_group.py) under two settings of ONE environment variable, same box:
#   tools/timeline_ab.sh LEGION_LDS_SMALL_BUCKETS 8 16 [extra bench.py args]   -> gpurun_out/timeline_ab_<var>.md
VAR=$1; A=$2; B=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
OUT=$R/gpurun_out/timeline_ab_$VAR.md; mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
for v in $A $B $A $B; do
  rm -rf /tmp/tl_ab
  env $VAR=$v timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_ab -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify --steps 8 --warmup 2 --min-seconds 0.3 "$@" > /tmp/tl_ab.json 2> /tmp/tl_ab.err < /dev/null
  echo "## $VAR=$v  (value $(python3 -c "import json;print(round(json.loads(open('/tmp/tl_ab.json').read().strip().splitlines()[-1])['value']/1e9,3))") G edges/s under the tracer)" >> $OUT
  python3 $R/tools/trace_group.py /tmp/tl_ab >> $OUT
done
cat $OUT
