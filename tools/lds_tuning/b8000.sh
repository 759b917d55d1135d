#!/bin/bash
# GPU box: the three forms of the first-touch state at Legion's default batch size (B = 8000), whole-job bench value
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
mkdir -p $R/gpurun_out/b8000
for F in direct lds; do
  for FO in "25,10" "15,10,5"; do
    LEGION_DEDUP=$F timeout -k 5 400 python3 $R/bench.py --batch 8000 --fanout $FO --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify $EXTRA 2> $R/gpurun_out/b8000/$F-$FO.err < /dev/null | tail -1 > $R/gpurun_out/b8000/$F-$FO.json
    python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/b8000/$F-$FO.json').read()); print('$F', '$FO', 'value', d['value']/1e9, 'ms', d['ms_per_step'], d['position_state']['form'], d['roofline']['frac'])"
  done
done
