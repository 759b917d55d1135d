#!/bin/bash
# whole-job A/B of library builds on one shape:  EXTRA="--batch 8000 --fanout 15,10,5" bash tools/value_ab.sh base v0 base v0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
for V in "$@"; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  echo -n "$V: "; bash $R/tools/gather_experiments.sh ab_$V -- $EXTRA
done
