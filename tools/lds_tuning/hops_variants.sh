#!/bin/bash
# dedup_hops.sh for several builds on the same box:  bash tools/lds_tuning/hops_variants.sh seqd v0 ...   (v0 = the library in place)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
for V in "$@"; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  echo "#### $V $EXTRA"; bash $R/tools/lds_tuning/dedup_hops.sh | grep "${FILTER:-.}"
done
