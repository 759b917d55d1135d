#!/bin/bash
# GPU box: per-kernel medians of one shape under each form:  EXTRA='--batch 8000 --fanout 15,10,5' FORMS='direct lds' bash tools/lds_tuning/trace_shape.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
for F in ${FORMS:-direct lds}; do
  export LEGION_DEDUP=$F
  rm -rf $R/gpurun_out/dedupx
  bash $R/tools/lds_tuning/kernel_times.sh > /dev/null 2>&1
  echo "######## $F"; cat $R/gpurun_out/dedupx/summary.txt
done
