"""Per-position average kernel durations of one launch group from a rocprofv3 --kernel-trace csv directory
(`bench.py --no-weave` on one stream: the op list repeats every 16 kernels).  Usage: python tools/trace_group.py <dir>"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "lg::" in r["Kernel_Name"] and r["Grid_Size_Y"] == "256"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.OrderedDict()
idx = [i for i, r in enumerate(rows) if "batch_generate" in r["Kernel_Name"]]
per = idx[1] - idx[0]
for gi in idx[2:-1]:
    for j in range(per):
        r = rows[gi + j]
        acc.setdefault((j, r["Kernel_Name"].split("(")[0][-30:]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(" ".join("%s=%.0f" % (n.split("::")[-1][:10], sum(v) / len(v)) for (j, n), v in acc.items()), "sum=%.0f" % sum(sum(v) / len(v) for v in acc.values()))
