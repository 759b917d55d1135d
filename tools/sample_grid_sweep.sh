#!/bin/bash
# The workgroup cap of the strided sampler grids (LegionTuning.sample_max_wg) at B = 8000 and at the headline, alternating, one box:
#   bash tools/sample_grid_sweep.sh   -> gpurun_out/sample_grid_sweep.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/sample_grid_sweep.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --steps 20 --warmup 5 > /dev/null 2>&1
one() { tag=$1; v=$2; shift; shift
  LEGION_SAMPLE_MAX_WG=$v timeout -k 5 300 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --min-seconds 0.5 --steps 20 --warmup 5 "$@" 2> /dev/null < /dev/null > /tmp/sg.json
  python3 - "$tag max_wg=$v" <<'PY' | tee -a $OUT
import json, sys
d = json.loads(open("/tmp/sg.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.3f G edges/s, gather %.3f, sampler-only %.2f G" % (d["value"] / 1e9, d["roofline"]["frac"], d["sampling_only"]["edges_per_sec"] / 1e9), flush=True)
PY
}
for r in 1 2; do for v in 4096 2048 8192 16384; do one b8000 $v --batch 8000; done; done
for r in 1 2; do for v in 4096 2048 8192; do one headline $v; done; done
