#!/bin/bash
# The light stream's priority (LegionTuning.weave_priority: -1 low, 0 equal) over several shapes, alternating, one box -> gpurun_out/priority_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/priority_ab.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --steps 20 --warmup 5 > /dev/null 2>&1
one() { tag=$1; v=$2; shift; shift
  LEGION_WEAVE_PRIORITY=$v timeout -k 5 400 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --min-seconds 0.5 "$@" 2> /dev/null < /dev/null > /tmp/pa.json
  python3 - "$tag priority=$v" <<'PY' | tee -a $OUT
import json, sys
d = json.loads(open("/tmp/pa.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.3f G edges/s, gather %.3f, sampler-only %.2f G, lanes %d" % (d["value"] / 1e9, d["roofline"]["frac"], d["sampling_only"]["edges_per_sec"] / 1e9, d["batches_per_step"]), flush=True)
PY
}
for r in 1 2; do for v in -1 0; do one b4096 $v --batch 4096 --steps 20 --warmup 5; done; done
for r in 1 2; do for v in -1 0; do one b2048_10_10 $v --batch 2048 --fanout 10,10 --steps 20 --warmup 5; done; done
for r in 1 2; do for v in -1 0; do one b8000_3hop $v --batch 8000 --fanout 15,10,5 --steps 16 --warmup 4; done; done
for r in 1 2; do for v in -1 0; do one b8000_d256 $v --batch 8000 --dim 256 --steps 16 --warmup 4; done; done
for r in 1 2; do for v in -1 0; do one b8000_25_10_10 $v --batch 8000 --fanout 25,10,10 --steps 16 --warmup 4; done; done
for r in 1 2; do for v in -1 0; do one d64 $v --dim 64 --steps 20 --warmup 5; done; done
