#!/bin/bash
# Counter evidence for the last-hop gather on the shapes the BASELINE configs use.
#   bash tools/gather_shapes_pmc.sh r03 [shape ...]      -> gpurun_out/gather_pmc_<round>/<shape>/g<k>/...
# One rocprofv3 --kernel-trace --pmc pass per counter group (never combined with other trace domains), each a
# short bench.py run of the shape; tools/gather_shapes_summary.py folds them into one table.
RND=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
OUT=$R/gpurun_out/gather_pmc_$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A SHAPES
SHAPES[headline]=""
SHAPES[d256_b8000]="--batch 8000 --dim 256"
SHAPES[3hop_15_10_5_b8000]="--batch 8000 --fanout 15,10,5"
SHAPES[products_d100]="--scale 21 --edge-factor 29 --dim 100"
SHAPES[d64]="--dim 64"
SHAPES[d602_rmat22]="--scale 22 --dim 602"
SHAPES[papers100m_size_3hop]="--nodes 111059956 --edges 1615685872 --batch 8000 --fanout 15,10,5 --group 8"
SHAPES[uk_union_size_d256]="--nodes 133633040 --edges 5507679822 --dim 256 --batch 8000 --group 8"
LIST="$@"
[ -z "$LIST" ] && LIST="headline d256_b8000 3hop_15_10_5_b8000 products_d100 d64 papers100m_size_3hop uk_union_size_d256"
GROUPS_PMC=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"
)
[ -n "$PMC_GROUPS" ] && IFS=';' read -ra GROUPS_PMC <<< "$PMC_GROUPS"
for shape in $LIST; do
  args=${SHAPES[$shape]}
  k=0
  for grp in "${GROUPS_PMC[@]}"; do
    k=$((k+1))
    tmp=/tmp/gpmc_${shape}_g$k
    rm -rf $tmp; mkdir -p $tmp
    t0=$(date +%s)
    timeout -k 5 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $tmp -- python3 $R/bench.py $args $EXTRA_BENCH_ARGS --steps 2 --warmup 1 --presc-steps 64 --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --min-seconds 0.01 > $tmp/bench.json 2> $tmp/err.txt < /dev/null
    rc=$?
    dst=$OUT/$shape/g$k
    mkdir -p $dst
    python3 $R/tools/gather_shapes_summary.py fold $tmp $dst "$grp"
    echo "$shape g$k ($grp) rc=$rc $(( $(date +%s) - t0 )) s"
  done
done
python3 $R/tools/gather_shapes_summary.py table $OUT > $OUT/table.md
cat $OUT/table.md
