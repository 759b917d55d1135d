#!/bin/bash
# Throughput of the server <-> trainer boundary in every hand-over mode, one data set (RMAT-$SCALE written once per batch size):
#   views  = whole launch groups into the lane arena, batches handed over as views (this build's ipc_service / native consumer)
#   slab   = the same server, trainer end without views: sampler phase in groups, one gather launch per batch into the pipe slot
#   gather = that hand-over forced for every trainer end (LEGION_RUNNER_HANDOVER=gather: no arena, lanes without feature buffers)
#   copy   = whole groups + one copy launch per batch into the pipe slot (LEGION_RUNNER_HANDOVER=copy)
# Usage (GPU box): tools/boundary_modes.sh [scale=22] [batches="1024,8000"] [train-batches=3000]  -> gpurun_out/boundary_modes_<scale>.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
SCALE=${1:-22}; BATCHES=${2:-1024,8000}; TB=${3:-3000}
OUT=$R/gpurun_out/boundary_modes_$SCALE.jsonl
mkdir -p $R/gpurun_out; : > $OUT
EXTRA=""; [ "$SCALE" -ge 25 ] && EXTRA="--no-features-file"
for consumer in python native; do
  for mode in "views:auto:" "slab:auto:--no-views" "gather:gather:" "copy:copy:--no-views"; do
    IFS=: read name handover flag <<< "$mode"
    echo "== $consumer $name" >&2
    python $R/tools/server_throughput.py --scale $SCALE --batch $BATCHES --dim 128 --train-batches $TB --consumer $consumer \
        --handover $handover $flag $EXTRA --watchdog 500 --cache-memory $((8<<30)) 2> $R/gpurun_out/boundary_modes_err.txt \
      | sed "s/^{/{\"mode\": \"$name\", \"consumer_kind\": \"$consumer\", /" >> $OUT || { tail -30 $R/gpurun_out/boundary_modes_err.txt >&2; exit 1; }
    grep -h "^runner " $R/gpurun_out/boundary_modes_err.txt >&2
  done
done
cat $OUT
