#!/bin/bash
# FETCH_SIZE and the L2 hit rate of the last hop's gather, warm (a group's own ids: 71 % repeats) against cold (ids that repeat nowhere):
# bench.py --cold-leg launches the same kernel 7 x alone and 7 x cold after the timed region (legion_pipeline_regather_last); the
# dispatches are told apart by their order.   bash tools/pmc_cold_vs_warm.sh  -> gpurun_out/r06/pmc_cold_vs_warm.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
OUT=$R/gpurun_out/r06/pmc_cold_vs_warm.txt
mkdir -p $R/gpurun_out/r06; : > $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_cw_$(echo $grp | cut -d' ' -f1); rm -rf $d; mkdir -p $d
  timeout -k 5 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/bench.py --steps 2 --warmup 1 --presc-steps 64 \
      --cpu-seconds 0 --no-verify --no-boundary --cold-leg --min-seconds 0.01 > $d/bench.json 2> $d/err.txt < /dev/null
  python3 - $d "$grp" >> $OUT <<'PY'
import csv, glob, sys, collections
d, grp = sys.argv[1], sys.argv[2].split()
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if "gather_kernel" in r["Kernel_Name"] and r["Kernel_Name"][:r["Kernel_Name"].rfind("(")].rstrip().endswith("true>")]
full = max(int(r["Grid_Size"]) for r in rows)
by = collections.OrderedDict()
for r in rows:
    if int(r["Grid_Size"]) != full: continue
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by, key=int)
last14 = ids[-14:]              # 7 x alone, then 7 x cold: the regather launches are the process's last dispatches of this kernel
for name, sel in (("in-group (timed steps)", ids[:-14]), ("alone (warm ids)", last14[:7]), ("cold (no id repeats)", last14[7:])):
    if not sel: continue
    avg = {c: sum(by[i].get(c, 0.0) for i in sel) / len(sel) for c in grp}
    print(" ".join(grp), "|", name, "| launches", len(sel), "|", "  ".join("%s %.1f" % (c, avg[c]) for c in grp))
PY
done
cat $OUT
