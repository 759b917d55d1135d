#!/bin/bash
# A/B of library builds on the gather: the cold-row probe + bench shapes.   bash tools/gather_ab.sh base v0 ...   (v0 = the library in place)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
for V in "$@"; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  echo "#### $V"
  python3 $R/tools/gather_probe.py 8000000 128 26 2>&1 | grep table
  python3 $R/tools/gather_probe.py 4000000 256 25 2>&1 | grep table
  python3 $R/tools/gather_probe.py 4000000 100 22 2>&1 | grep table
  bash $R/tools/gather_experiments.sh $V -- -- --batch 8000 --dim 256 -- --batch 8000 --fanout 15,10,5 -- --dim 64 -- --scale 21 --edge-factor 29 --dim 100
done
