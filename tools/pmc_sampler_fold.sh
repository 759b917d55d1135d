#!/bin/bash
# Sampler-chain counters of the bench's default workload, one counter group per run (kernel-trace only beside the counters), folded
# into one table per kernel and grid:   tools/pmc_sampler_fold.sh [tag] [extra bench.py args...]  -> gpurun_out/pmc_<tag>/table.md
TAG=${1:-sampler}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  mkdir -p $OUT/g$i
  timeout -k 5 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 2 --warmup 1 --presc-steps 64 \
      --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --min-seconds 0.01 "$@" > $OUT/g$i/bench.json 2> $OUT/g$i/err.txt < /dev/null
  echo "group $i ($grp) rc=$?"
  find $OUT/g$i -name "*agent_info.csv" -delete
done
python3 $R/tools/pmc_fold.py $OUT $OUT/table.md
cp $OUT/g1/bench.json $OUT/bench_under_counters.json 2>/dev/null
rm -rf $OUT/g[0-9]*            # the raw per-dispatch csv files are tens of MB: only the folded table travels back
