#!/bin/bash
# LegionTuning.feature_pitch on the products shape (D = 100: 400-byte rows): dense rows against rows padded to whole 128-byte lines.
# Per setting: the bench line (value, gather fraction) and the HBM traffic of the last hop's gather (separate FETCH_SIZE / WRITE_SIZE
# passes, gfx950 corrections in tools/pmc_summary.py).   tools/gather_pitch.sh  -> gpurun_out/gather_pitch/summary.md
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/gather_pitch; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPE="--scale 21 --edge-factor 29 --dim 100 --cpu-seconds 0 --no-boundary --no-overlap-leg"
echo "| feature_pitch | G edges/s | gather frac of 8 TB/s (HIP events) | read MB / launch | written MB / launch | traffic / algorithmic |" > $OUT/summary.md
echo "|---|---|---|---|---|---|" >> $OUT/summary.md
for pitch in dense aligned; do
  export LEGION_FEATURE_PITCH=$pitch
  timeout -k 5 400 python3 $R/bench.py $SHAPE > $OUT/bench_$pitch.json 2> $OUT/bench_$pitch.err < /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/gp_$c; mkdir -p /tmp/gp_$c
    timeout -k 5 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/gp_$c -- python3 $R/bench.py $SHAPE --steps 4 --warmup 2 --presc-steps 64 --no-verify --min-seconds 0.01 > /tmp/gp_$c/bench.json 2> /tmp/gp_$c/err.txt < /dev/null
  done
  python3 $R/tools/pmc_summary.py /tmp/gp_FETCH_SIZE /tmp/gp_WRITE_SIZE $OUT/pmc_$pitch.json > /dev/null
  python3 - $OUT/bench_$pitch.json $OUT/pmc_$pitch.json $pitch >> $OUT/summary.md <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p = json.load(open(sys.argv[2]))
print("| %s | %.3f | %.3f | %.1f | %.1f | %.3f |" % (sys.argv[3], b["value"] / 1e9, b["roofline"]["frac"], p["read_bytes_per_launch_corrected"] / 1e6,
      p["write_bytes_per_launch"] / 1e6, p["traffic_over_algorithmic"]))
PY
done
cat $OUT/summary.md
