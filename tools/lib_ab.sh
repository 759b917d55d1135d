#!/bin/bash
# Whole-job bench lines for several builds of the library, alternating, same box (no profiler):
#   bash tools/lib_ab.sh "v0 name ..." <tag> <rounds> [bench.py args...]      v0 = the library in place, name = tools/lds_tuning/variants/<name>/
VARS=$1; TAG=$2; ROUNDS=$3; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/lib_ab_$TAG.txt; : >> $OUT
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    LIBV=""; [ $v != v0 ] && LIBV=$R/tools/lds_tuning/variants/$v/liblegion_hip.so
    LEGION_HIP_LIB=$LIBV timeout -k 5 400 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --min-seconds 0.5 "$@" > /tmp/ab.json 2> /tmp/ab.err < /dev/null
    python3 - "$TAG $v" <<'PY' | tee -a $OUT
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "gather frac %.3f" % d["roofline"]["frac"],
          "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("/tmp/ab.err").read()[-600:])
PY
  done
done
