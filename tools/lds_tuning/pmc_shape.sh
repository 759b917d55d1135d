#!/bin/bash
# GPU box: PMC view of the sampler kernels at another shape:  EXTRA='--batch 8000 --fanout 15,10,5' bash tools/lds_tuning/pmc_shape.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.."; pwd)}
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_ATOMIC_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum"; do
  out=/tmp/pmcs_$RANDOM; rm -rf $out; mkdir -p $out
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $R/bench.py $EXTRA --steps 2 --warmup 1 --presc-steps 16 --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --no-weave --min-seconds 0.01 > $out/bench.json 2> $out/err.txt < /dev/null
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')
if not fs: print("no csv", open(sys.argv[1] + "/err.txt").read()[-400:]); sys.exit()
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'sample_kernel' not in n and 'dedup_l' not in n: continue
    k = (n.split('(')[0][:44], r['Grid_Size'], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for k, v in sorted(acc.items()):
    print(k, "avg %.0f" % (v[0] / v[1]), "n", v[1])
PY
done
