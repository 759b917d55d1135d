#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box:  bash tools/profile_round.sh r02
# Output (small files only) under gpurun_out/prof_<round>/ ; copy into profiles/<round>/ afterwards.
RND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
OUT=$R/gpurun_out/prof_$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 0. a minute of load first (the driver's bench also follows its GPU test run): about every second fresh box of this pool runs its first
#    50-70 s of GPU work 2-10 % slow -- sampler and gathers alike -- and then settles (profiles/r05/README.md); WARM=0 skips this
if [ "${WARM:-1}" != "0" ]; then
  for i in 1 2 3; do timeout -k 5 300 python3 $R/bench.py --cpu-seconds 0 --no-boundary --no-verify --steps 20 --warmup 5 2> /dev/null < /dev/null | tail -1 | cut -c1-120; done > $OUT/warm_up_runs.txt
fi
# 1. the bench line as the driver runs it, and at its defaults
timeout -k 5 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver_args.err < /dev/null | tail -1 > $OUT/bench_driver_args.json
timeout -k 5 600 python3 $R/bench.py 2> $OUT/bench_default.err < /dev/null | tail -1 > $OUT/bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no boundary leg: that is another process)
rm -rf /tmp/prof_stats
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 > $OUT/bench_default_under_rocprof.json 2> /tmp/prof_stats.err < /dev/null
cp /tmp/prof_stats/*/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv
cp /tmp/prof_stats/*/*domain_stats.csv $OUT/bench_default_domain_stats.csv
python3 - /tmp/prof_stats $OUT/bench_default_kernel_trace_by_grid.csv <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if not (n.startswith('lg::') or 'lg::' in n): continue
    key = (n[:n.rfind('(')].replace('float __vector(4)', 'float4').replace(' ', '')[:64], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r.get('VGPR_Count', r.get('Arch_VGPR_Count', '')), r.get('LDS_Block_Size', ''))
    acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(sys.argv[2], 'w') as o:
    o.write("kernel,grid_x_threads,grid_y,workgroup,vgpr,lds_bytes,launches,avg_us,min_us,max_us,total_us\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        o.write(",".join(k) + f",{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v):.1f}\n")
PY
# 3. HBM traffic of the dominant kernel: separate --pmc passes (kernel-trace only), gfx950 corrections in pmc_summary.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; mkdir -p /tmp/pmc_$c
  timeout -k 5 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 4 --warmup 2 --presc-steps 64 --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --min-seconds 0.01 > /tmp/pmc_$c/bench.json 2> /tmp/pmc_$c/err.txt < /dev/null
done
python3 $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/pmc_gather_kernel.json > /dev/null
# 4. the timeline of one launch group under the weave (kernel trace + roctx group ranges), folded by tools/trace_group.py
rm -rf /tmp/prof_tl
timeout -k 5 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/prof_tl -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify --steps 8 --warmup 2 --min-seconds 0.3 > /dev/null 2> /tmp/prof_tl.err < /dev/null
python3 $R/tools/trace_group.py /tmp/prof_tl $OUT/group_timeline.md > /dev/null
cut -c1-300 $OUT/bench_default.json
head -14 $OUT/bench_default_kernel_trace_by_grid.csv
cat $OUT/pmc_gather_kernel.json
