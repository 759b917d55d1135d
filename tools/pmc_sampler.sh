#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) for the sampler kernels; output under gpurun_out/pmc_s/<group>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "MemUnitStalled LDSBankConflict" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_REQ_sum"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_s/g$i
  mkdir -p $out
  echo "$grp" > $out/group.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $R/bench.py --group 64 --steps 128 --warmup 64 --presc-steps 64 --cpu-seconds 0 --no-verify > $out/bench.json 2> $out/err.txt
  echo "group $i rc=$?"
  # keep only the folded summary (the raw csv is large)
  python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
fs = glob.glob(d + '/*/*counter_collection.csv')
if not fs: print("no csv"); sys.exit()
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    k = (r['Kernel_Name'].split('(')[0][:40], r['Grid_Size'], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
with open(d + '/summary.csv', 'w') as f:
    f.write("kernel,grid,counter,avg,launches\n")
    for k, v in sorted(acc.items()):
        f.write(f"{k[0]},{k[1]},{k[2]},{v[0]/v[1]:.1f},{v[1]}\n")
import os
for p in fs + glob.glob(d + '/*/*kernel_trace.csv') + glob.glob(d + '/*/*agent_info.csv'):
    os.remove(p)
PY
done
