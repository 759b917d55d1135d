#!/bin/bash
# Boundary throughput (sampling_server -> ipc_service consumer) at RMAT-24, B = 1024 and 8000, python and native consumers:
#   bash tools/boundary_ab.sh <tag>     -> gpurun_out/boundary_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/boundary_$1.txt; : > $OUT
for c in python native; do
  timeout -k 5 500 python3 $R/tools/server_throughput.py --scale 24 --batch 1024,8000 --train-batches 3000 --no-features-file --consumer $c --epochs 3 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('$c', 'B', d['batch'], '%.0f batches/s' % d['batches_per_sec'], '%.3f G edges/s' % (d['edges_per_sec'] / 1e9))" >> $OUT
done
cat $OUT
