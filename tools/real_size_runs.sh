#!/bin/bash
# Bench lines at the reference's REAL data-set sizes on one MI355X (legion_server.py:41-88):  bash tools/real_size_runs.sh r03
#   uk-union   N = 133 633 040, E = 5 507 679 822, D = 256, B = 8000, [25,10]   (configs[3]'s data set; all resident: 160 GB)
#   papers100M N = 111 059 956, E = 1 615 685 872, D = 128, B = 8000, [15,10,5], CSR + features in pinned host memory (configs[2])
RND=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}; OUT=$R/gpurun_out/real_$RND; mkdir -p $OUT
# GROUP: lanes per launch group.  Rounds 3-4 ran these shapes with 8 (a cautious guess at what fits beside 160-200 GB of tables); since round 5
# the default is bench.py's own rule -- 524288 / B lanes, halved while the lanes of the groups in flight would take more than 0.7 of the
# HBM the tables left free: 64 lanes at every one of these shapes -- and GROUP=8 reproduces the old lines.
COMMON="--cpu-seconds 0 --no-boundary --presc-steps 704 --steps 8 --warmup 2 ${GROUP:+--group $GROUP}"
run() { name=$1; shift; ( time timeout -k 5 1200 python3 $R/bench.py $COMMON "$@" > $OUT/$name.json 2> $OUT/$name.err < /dev/null ) 2>&1 | grep real
  python3 - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f G edges/s" % (d["value"] / 1e9), "ms/step %.3f" % d["ms_per_step"], "batches/step", d["batches_per_step"], "gather frac %.3f" % d["roofline"]["frac"],
          "rows/launch %.0f" % d["roofline"]["rows_per_launch"], "sampler-only %.2f G" % (d["sampling_only"]["edges_per_sec"] / 1e9),
          "cache rows", d["config"]["feature_cache_rows"], "hit rate %.3f" % d["feature_cache_hit_rate"], d.get("miss_path", {}).get("pcie_feature_GBps"))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
if [ "${PART:-all}" != "rest" ]; then
run uk_union_size_d256_b8000 --nodes 133633040 --edges 5507679822 --dim 256 --batch 8000          # (column slots: auto takes the 44 GB copy since round 4)
LEGION_COL_SLOTS=0 run uk_union_size_d256_b8000_no_column_slots --nodes 133633040 --edges 5507679822 --dim 256 --batch 8000
fi
[ "${PART:-all}" = "uk" ] && exit 0
run papers100m_size_3hop_pinned --nodes 111059956 --edges 1615685872 --dim 128 --batch 8000 --fanout 15,10,5 --placement pinned --link-counters smi
run papers100m_size_3hop_hbm --nodes 111059956 --edges 1615685872 --dim 128 --batch 8000 --fanout 15,10,5
# configs[4]'s graph on one GPU: RMAT-28 (N = 2^28, edge factor 4), [15,10,5], B = 8000 -- with D = 128 (the 256-wide table of 2^28 rows
# is 275 GB and exists only striped over eight GPUs)
run rmat28_ef4_d128_3hop --scale 28 --edge-factor 4 --dim 128 --batch 8000 --fanout 15,10,5
