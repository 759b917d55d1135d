#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box:  bash tools/profile_round.sh r01
# Output (small files only) under gpurun_out/prof_<round>/ ; copy into profiles/<round>/ afterwards.
RND=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py > $OUT/bench_default_under_rocprof.json 2> /tmp/prof_stats.err
cp /tmp/prof_stats/*/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv
cp /tmp/prof_stats/*/*domain_stats.csv $OUT/bench_default_domain_stats.csv
python3 - /tmp/prof_stats $OUT/bench_default_kernel_trace_by_grid.csv <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if not (n.startswith('lg::') or 'lg::' in n): continue
    key = (n.split('(')[0][:48], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r.get('VGPR_Count', r.get('Arch_VGPR_Count', '')), r.get('LDS_Block_Size', ''))
    acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(sys.argv[2], 'w') as o:
    o.write("kernel,grid_x_threads,grid_y,workgroup,vgpr,lds_bytes,launches,avg_us,min_us,max_us,total_us\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        o.write(",".join(k) + f",{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v):.1f}\n")
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; mkdir -p /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 256 --warmup 128 --presc-steps 64 --cpu-seconds 0 --no-verify > /tmp/pmc_$c/bench.json 2> /tmp/pmc_$c/err.txt
done
python3 $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/pmc_gather_kernel.json > /dev/null
cat $OUT/bench_default.json | cut -c1-400
head -12 $OUT/bench_default_kernel_trace_by_grid.csv
cat $OUT/pmc_gather_kernel.json
