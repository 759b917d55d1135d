#!/bin/bash
# GPU box: partition-tile size of the large-batch LDS form (LEGION_LDS_PART_WG = workgroups a sampling launch should at least have)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
for FO in "25,10" "15,10,5"; do
  for W in ${WLIST:-2048 8192 16384 32768}; do
    LEGION_LDS_PART_WG=$W timeout -k 5 400 python3 $R/bench.py --batch 8000 --fanout $FO --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify $EXTRA 2> /dev/null < /dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$FO', 'part_wg $W', 'value', round(d['value']/1e9,3), 'ms', round(d['ms_per_step'],4))"
  done
done
