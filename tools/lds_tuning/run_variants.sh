#!/bin/bash
# GPU box:  VLIST='v0 v16 ...' bash tools/lds_tuning/run_variants.sh   (v0 = the library in place)
# Selects each variant library through LEGION_HIP_LIB (legion_amd/lib.py) -- the library in place is never touched --
# and runs kernel_times.sh and the default bench with it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
mkdir -p $R/gpurun_out/dedupx
for V in $VLIST; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  echo "######## $V" >> $R/gpurun_out/dedupx/summary.txt
  bash $R/tools/lds_tuning/kernel_times.sh > /dev/null 2>&1
  mv $R/gpurun_out/dedupx/x0.json $R/gpurun_out/dedupx/$V.json
  # the weave default too
  timeout -k 5 300 python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify $EXTRA 2>/dev/null < /dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('weave value', d['value']/1e9, 'ms', d['ms_per_step'])" >> $R/gpurun_out/dedupx/summary.txt
done
unset LEGION_HIP_LIB
cat $R/gpurun_out/dedupx/summary.txt
