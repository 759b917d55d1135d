#!/bin/bash
# One-line summaries of bench.py runs that vary only the gather: bash tools/gather_experiments.sh <tag> -- <bench args> [-- <bench args> ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/gather_exp; mkdir -p $OUT
TAG=$1; shift; shift
i=0
args=()
run() {
  i=$((i+1))
  timeout -k 5 400 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-overlap-leg --no-verify "${args[@]}" > $OUT/${TAG}_$i.json 2> $OUT/${TAG}_$i.err < /dev/null
  python3 - $OUT/${TAG}_$i.json "${args[*]}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-60s value %.3f G  ms/step %.4f  gather frac %.3f  us %.0f  rows %.0f  sampler-only %.2f G" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], r["frac"], r["avg_launch_us"], r["rows_per_launch"], d["sampling_only"]["edges_per_sec"] / 1e9))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for a in "$@"; do
  if [ "$a" = "--" ]; then run; args=(); else args+=("$a"); fi
done
run
