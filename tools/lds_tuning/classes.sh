#!/bin/bash
# GPU box: bucket counts of the medium / large classes (variant libraries built by build_variant.sh) at B = 8000;
# variants are selected through LEGION_HIP_LIB, the library in place is never touched
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
for V in $VLIST; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  for FO in "25,10" "15,10,5" "25,10,10"; do
    timeout -k 5 400 python3 $R/bench.py --batch 8000 --fanout $FO --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify 2> /dev/null < /dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', '$FO', 'value', round(d['value']/1e9,3), 'ms', round(d['ms_per_step'],4))"
  done
done
unset LEGION_HIP_LIB
