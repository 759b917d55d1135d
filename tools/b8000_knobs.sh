#!/bin/bash
# B = 8000 whole-job lines under a few arrangements, same box:  bash tools/b8000_knobs.sh  -> gpurun_out/b8000_knobs.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/b8000_knobs.txt; : > $OUT
run() { tag=$1; shift
  timeout -k 5 600 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify "$@" > /tmp/k.json 2> /tmp/k.err < /dev/null
  python3 - /tmp/k.json "$tag" <<'PY' | tee -a $OUT
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "batches/step", d["batches_per_step"], "gather frac %.3f" % d["roofline"]["frac"],
          "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open("/tmp/k.err").read()[-600:])
PY
}
for r in 1 2; do
run "b8000 slots2 r$r" --batch 8000 --slots 2
run "b8000 slots3 r$r" --batch 8000 --slots 3
LEGION_WEAVE_PRIORITY=0 run "b8000 slots2 equal-priority r$r" --batch 8000 --slots 2
run "b8000 slots3 group32 r$r" --batch 8000 --slots 3 --group 32
done
run "headline slots3" --slots 3
run "b8000 3hop slots2" --batch 8000 --fanout 15,10,5 --slots 2
run "b8000 3hop slots3" --batch 8000 --fanout 15,10,5 --slots 3
