#!/bin/bash
# A/B of variant builds in which the gather loads rows that are unlikely to repeat inside a launch (hotness rank >= K, or table misses)
# with the non-temporal hint (they then bypass the Infinity Cache and leave it to the rows that do repeat), against the product library.
# Variant builds: tools/experiments/nt_cold.patch (tools/experiments/README.md).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06/nt_cold_ab.txt
mkdir -p $R/gpurun_out/r06; : > $OUT
cd /tmp
for rep in 1 2; do
for v in base ntcold262144 ntcold1048576 ntcold4194304 ntcold1073741824; do
  if [ $v = base ]; then L=$R/legion_amd/liblegion_hip.so; else L=$R/tools/lds_tuning/variants/$v/liblegion_hip.so; fi
  for shape in "" "--batch 8000"; do
  LEGION_HIP_LIB=$L timeout -k 5 300 python3 $R/bench.py --no-boundary --cold-leg --cpu-seconds 0 --steps 20 --warmup 5 $shape 2> /tmp/nc.err < /dev/null | tail -1 > /tmp/nc.json
  python3 - $v "$shape" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/nc.json"))
    r = d["roofline"]
    print(sys.argv[1], sys.argv[2] or "headline", "value %.3f G" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "gather frac %.4f us %.1f" % (r["frac"], r["avg_launch_us"]),
          "alone %.4f" % ((r.get("alone") or {}).get("frac") or 0), "cold %.4f" % ((r.get("cold") or {}).get("frac") or 0))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed", e)
PY
  done
done
done
cat $OUT
