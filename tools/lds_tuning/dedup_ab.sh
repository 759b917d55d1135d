#!/bin/bash
# A/B of sampler-chain builds on one box: parity tests with the variant and the group timeline (tools/trace_group.py).
#   [PARITY="base x"] [EXTRA=--no-weave] tools/lds_tuning/dedup_ab.sh x y      (variants v_<name> built by
#   tools/lds_tuning/build_variant.sh; variants/v_<name>/env, if present, holds `export` lines for that variant's runs)
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
