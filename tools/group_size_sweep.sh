#!/bin/bash
# Real-size shapes at several launch-group sizes (memory permitting):  bash tools/group_size_sweep.sh "8 16 32 0" [shape ...]    (0 = bench.py's own rule)
GROUPS_=${1:-"8 32 0"}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/group_sweep; mkdir -p $OUT
SHAPES=${@:-"uk rmat28"}
for s in $SHAPES; do
  case $s in
    uk) ARGS="--nodes 133633040 --edges 5507679822 --dim 256 --batch 8000" ;;
    papers) ARGS="--nodes 111059956 --edges 1615685872 --dim 128 --batch 8000 --fanout 15,10,5" ;;
    rmat28) ARGS="--scale 28 --edge-factor 4 --dim 128 --batch 8000 --fanout 15,10,5" ;;
  esac
  for g in $GROUPS_; do
    timeout -k 5 900 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-overlap-leg --presc-steps 64 --steps 8 --warmup 2 --group $g $ARGS > $OUT/${s}_g$g.json 2> $OUT/${s}_g$g.err < /dev/null
    python3 - $OUT/${s}_g$g.json ${s}_g$g <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "batches/step", d["batches_per_step"], "gather frac %.3f" % d["roofline"]["frac"],
          "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
  done
done
