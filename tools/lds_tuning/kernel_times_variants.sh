#!/bin/bash
# kernel_times.sh for several builds of the library:  bash tools/lds_tuning/kernel_times_variants.sh old v0 mw6   (v0 = the library in place)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
rm -f $R/gpurun_out/dedupx/summary.txt
for V in "$@"; do
  if [ $V = v0 ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$R/tools/lds_tuning/variants/$V/liblegion_hip.so; fi
  mkdir -p $R/gpurun_out/dedupx; echo "#### variant $V $EXTRA" >> $R/gpurun_out/dedupx/summary.txt
  bash $R/tools/lds_tuning/kernel_times.sh > /dev/null
done
unset LEGION_HIP_LIB
cat $R/gpurun_out/dedupx/summary.txt
