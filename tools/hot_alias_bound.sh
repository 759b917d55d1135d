#!/bin/bash
# Upper bound of reading every hot row once: variant builds from tools/experiments/hot_alias.patch (tools/experiments/README.md) against the product library.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06/hot_alias_bound.txt
mkdir -p $R/gpurun_out/r06
: > $OUT
cd /tmp
for rep in 1 2; do
for v in base hot65536 hot262144 hot1048576 hot16777216; do
  if [ $v = base ]; then L=$R/legion_amd/liblegion_hip.so; else L=$R/tools/lds_tuning/variants/$v/liblegion_hip.so; fi
  LEGION_HIP_LIB=$L timeout -k 5 300 python3 $R/bench.py --no-boundary --cold-leg --cpu-seconds 0 --no-verify --steps 20 --warmup 5 2> /tmp/ha.err < /dev/null | tail -1 > /tmp/ha.json
  python3 - $v >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ha.json"))
    r = d["roofline"]
    print(sys.argv[1], "value %.3f G" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "gather frac %.4f us %.1f" % (r["frac"], r["avg_launch_us"]),
          "alone %.4f" % ((r.get("alone") or {}).get("frac") or 0), "cold %.4f" % ((r.get("cold") or {}).get("frac") or 0))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
done
cat $OUT
