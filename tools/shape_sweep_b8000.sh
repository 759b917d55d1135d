#!/bin/bash
# The B = 8000 lines of tools/shape_sweep.sh again, behind a minute of load: about every second fresh box of this pool runs its first
# 50-70 s of GPU work 6-10 % slow (sampler and gathers alike; within a run the timed regions agree to 0.3 %), then settles.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/shapes; mkdir -p $OUT
for i in 1 2 3; do timeout -k 5 300 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --steps 20 --warmup 5 2> /dev/null < /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warm-up run %.3f G edges/s, gather frac %.3f' % (d['value']/1e9, d['roofline']['frac']))"; done
run() { name=$1; shift; timeout -k 5 400 python3 $R/bench.py --cpu-seconds 0 --no-boundary "$@" > $OUT/$name.json 2> $OUT/$name.err < /dev/null
  python3 - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.4f" % d["ms_per_step"], "batches/step", d["batches_per_step"],
          "gather frac %.3f" % d["roofline"]["frac"],
          "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9), "buckets", d["first_touch_state"].get("lds_buckets_per_lane"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run b8000_d128 --batch 8000
run b8000_d256 --batch 8000 --dim 256
run b8000_3hop_15_10_5 --batch 8000 --fanout 15,10,5
