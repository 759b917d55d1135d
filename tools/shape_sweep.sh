#!/bin/bash
# Other workload shapes on one MI355X: one bench line each under gpurun_out/shapes/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/shapes; mkdir -p $OUT
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
run b8000_3hop_25_10_10 --batch 8000 --fanout 25,10,10
run d64 --dim 64
run products_shape --scale 21 --edge-factor 29 --dim 100
run fanout_10_10_b2048 --batch 2048 --fanout 10,10
run d602_rmat22 --scale 22 --dim 602
run rmat27_ef8_d128 --scale 27 --edge-factor 8
