#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) for the sampler kernels, once per form of the position state:
#   bash tools/pmc_sampler.sh direct|table|lds [--scramble]     -> gpurun_out/pmc_s_<form>[_scr]/pmc_sampler_kernels.csv
FORM=${1:-direct}
SCR=$2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
TAG=pmc_s_$FORM$( [ -n "$SCR" ] && echo _scr )
i=0
rm -f $R/gpurun_out/$TAG/pmc_sampler_kernels.csv
mkdir -p $R/gpurun_out/$TAG
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  out=/tmp/$TAG/g$i
  rm -rf $out; mkdir -p $out
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $R/bench.py --dedup $FORM $SCR --steps 2 --warmup 1 --presc-steps 64 --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --min-seconds 0.01 > $out/bench.json 2> $out/err.txt < /dev/null
  echo "group $i ($grp) rc=$?"
  python3 - "$out" "$R/gpurun_out/$TAG/pmc_sampler_kernels.csv" "$FORM$SCR" <<'PY'
import csv, glob, sys, collections, os
d, dst, form = sys.argv[1], sys.argv[2], sys.argv[3]
fs = glob.glob(d + '/*/*counter_collection.csv')
if not fs: print("no csv"); sys.exit()
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'lg::' not in n: continue
    k = (n.split('(')[0][:48], r['Grid_Size'], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
new = not os.path.exists(dst)
with open(dst, 'a') as f:
    if new: f.write("position_state,kernel,grid_threads,counter,avg_per_launch,launches\n")
    for k, v in sorted(acc.items()):
        f.write(f"{form},{k[0]},{k[1]},{k[2]},{v[0]/v[1]:.1f},{v[1]}\n")
PY
done
wc -l $R/gpurun_out/$TAG/pmc_sampler_kernels.csv
