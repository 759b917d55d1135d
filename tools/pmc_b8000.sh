#!/bin/bash
# Sampler-chain counters of the B = 8000 shapes (BASELINE configs[2]-[4] all run B = 8000): seven rocprofv3 --pmc passes per shape, folded by
# tools/pmc_fold.py.   bash tools/pmc_b8000.sh <tag> [shape ...]   -> gpurun_out/pmc_<tag>_<shape>/table.md
TAG=${1:-r05}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
SHAPES=${@:-"b8000_d128 rmat28_ef4_d128_3hop uk_union_size_d256_b8000"}
for s in $SHAPES; do
  case $s in
    b8000_d128) ARGS="--batch 8000" ;;
    b8000_3hop) ARGS="--batch 8000 --fanout 15,10,5" ;;
    rmat28_ef4_d128_3hop) ARGS="--scale 28 --edge-factor 4 --dim 128 --batch 8000 --fanout 15,10,5 --group 8" ;;
    uk_union_size_d256_b8000) ARGS="--nodes 133633040 --edges 5507679822 --dim 256 --batch 8000 --group 8" ;;
    papers100m_size_3hop_hbm) ARGS="--nodes 111059956 --edges 1615685872 --dim 128 --batch 8000 --fanout 15,10,5 --group 8" ;;
    *) echo "unknown shape $s"; continue ;;
  esac
  echo "== $s: $ARGS"
  ( time bash $R/tools/pmc_sampler_fold.sh ${TAG}_$s $ARGS ) 2>&1 | grep -v "^|" | tail -12
done
