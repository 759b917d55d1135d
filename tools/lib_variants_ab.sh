#!/bin/bash
# Whole-job bench lines + the last hop's de-duplication time for several builds of the library (tools/lds_tuning/variants/<name>/, built with
# `python -m legion_amd.build --variant <name> -D...`), same box:  bash tools/lib_variants_ab.sh "v0 a b" <tag> [bench.py args...]   (v0 = in place)
VARS=$1; TAG=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/variants_$TAG.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  LIBV=""; [ $v != v0 ] && LIBV=$R/tools/lds_tuning/variants/$v/liblegion_hip.so
  rm -rf /tmp/tlv
  LEGION_HIP_LIB=$LIBV timeout -k 5 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/tlv -- python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --steps 8 --warmup 2 --min-seconds 0.3 "$@" > /tmp/lv.json 2> /tmp/lv.err < /dev/null
  python3 $R/tools/trace_group.py /tmp/tlv /tmp/lv.md > /dev/null 2>&1
  python3 - /tmp/lv.json "$TAG $v" /tmp/lv.md <<'PY' | tee -a $OUT
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    tl = [ln for ln in open(sys.argv[3]).read().splitlines() if "dedup" in ln or ln.startswith("step")]
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "gather frac %.3f" % d["roofline"]["frac"], "|", " ".join(t[:110] for t in tl), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open("/tmp/lv.err").read()[-600:])
PY
done
