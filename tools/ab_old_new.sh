#!/bin/bash
# Same-box A/B of whole-job bench lines: the round-4 tree (.old_r04/, built in place, git-ignored) against the tree in place, alternating.
#   bash tools/ab_old_new.sh <rounds> <tag> [bench.py args...]     -> gpurun_out/ab_<tag>.txt
ROUNDS=$1; TAG=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/ab_$TAG.txt; : > $OUT
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "batches/step", d["batches_per_step"], "gather frac %.3f" % d["roofline"]["frac"],
          "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
for r in $(seq 1 $ROUNDS); do
  for side in old new; do
    D=$R; [ $side = old ] && D=$R/.old_r04
    ( cd $D && timeout -k 5 600 python3 bench.py --cpu-seconds 0 --no-boundary --no-overlap-leg --no-verify "$@" > /tmp/ab_$side.json 2> /tmp/ab_$side.err < /dev/null )
    line /tmp/ab_$side.json "$TAG $side r$r" | tee -a $OUT
  done
done
