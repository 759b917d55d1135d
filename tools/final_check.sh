#!/bin/bash
# The round's closing check on the GPU box: the whole GPU suite, then a short bench line with its boundary leg.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log; tail -3 gpurun_out/t_all.log
timeout -k 10 280 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-overlap-leg --boundary-batches 8000 > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"].get("rocprofv3_avg_launch_us"), d["roofline"].get("rocprofv3_source"))
b = d["boundary"]
print([(l["mode"], l["batch"], round(l["batches_per_sec"])) for l in b.get("by_batch_size", []) + b.get("slab_only_trainer", {}).get("by_batch_size", [])], b.get("error"))
PY
