#!/bin/bash
# A/B of dedup_lds_kernel builds on one box: parity tests with the variant, its stamps build's phase table, and the group timeline.
#   tools/lds_tuning/dedup_ab.sh all c6 fix spec      (variants v_<name> and s_<name> built by tools/lds_tuning/build_variant.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
V=$R/tools/lds_tuning/variants
OUT=$R/gpurun_out/dedup_ab; mkdir -p $OUT; : > $OUT/summary.txt
BA="--no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify --steps 8 --warmup 2 --min-seconds 0.3"
cd $R
for n in ${PARITY-base "$@"}; do
  echo "=== parity with v_$n" >> $OUT/summary.txt
  if [ $n = base ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$V/v_$n/liblegion_hip.so; fi
  unset LEGION_LDS_SMALL_BUCKETS; [ -f $V/v_$n/env ] && . $V/v_$n/env
  timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_pipeline.py -x -q -m gpu > $OUT/parity_$n.log 2>&1
  rc=$?
  tail -2 $OUT/parity_$n.log >> $OUT/summary.txt
  if [ $rc != 0 ]; then echo "parity FAILED for v_$n" >> $OUT/summary.txt; cat $OUT/summary.txt; exit 1; fi
done
unset LEGION_HIP_LIB LEGION_LDS_SMALL_BUCKETS
for n in ${STAMPS-base "$@"}; do
  echo "=== stamps s_$n" >> $OUT/summary.txt
  LEGION_HIP_LIB=$V/s_$n/liblegion_hip.so timeout -k 5 250 python tools/lds_tuning/dedup_stamps.py $BA $EXTRA > $OUT/stamps_$n.json 2> $OUT/stamps_$n.err
  grep -A9 "^hop 2" $OUT/stamps_$n.err >> $OUT/summary.txt
done
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for n in base "$@"; do
  rm -rf /tmp/tl_ab
  if [ $n = base ]; then unset LEGION_HIP_LIB; else export LEGION_HIP_LIB=$V/v_$n/liblegion_hip.so; fi
  unset LEGION_LDS_SMALL_BUCKETS; [ -f $V/v_$n/env ] && . $V/v_$n/env        # (a variant's run-time settings: export lines)
  timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_ab -- python3 $R/bench.py $BA $EXTRA > /tmp/tl_ab.json 2> /tmp/tl_ab.err < /dev/null
  echo "## $n  (value $(python3 -c "import json;print(round(json.loads(open('/tmp/tl_ab.json').read().strip().splitlines()[-1])['value']/1e9,3))") G edges/s under the tracer)" >> $OUT/summary.txt
  python3 $R/tools/trace_group.py /tmp/tl_ab | grep -E "dedup|compact_kernel<true>|sample_kernel|^step" | cut -c1-170 >> $OUT/summary.txt
done
done
cat $OUT/summary.txt
