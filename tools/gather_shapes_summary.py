"""Folds the per-group rocprofv3 --pmc passes of tools/gather_shapes_pmc.sh.

    fold <rocprof dir> <dst dir> "<counter group>"   one pass -> <dst>/fold.json (last-hop gather launches only)
    table <root>                                     every <root>/<shape>/g*/fold.json -> a markdown table

gfx950 notes (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE tallies the 128-byte requests
of a wide stream at 64 B and is doubled here; the read bytes are also rebuilt from the request-size counters
(TCC_EA0_RDREQ_{32B,64B,128B}) where the part exposes them, which needs no calibration."""
import collections
import csv
import glob
import json
import os
import sys


def last_op(n):
    """gather_kernel<..., true>(...): the instance launched for a batch's last op (kernels_gather.hip LASTOP)."""
    return "gather_kernel" in n and n[:n.rfind("(")].rstrip().endswith("true>")


def fold(src, dst, group):
    os.makedirs(dst, exist_ok=True)
    out = {"group": group.split(), "counters": {}, "launches": 0}
    cc = glob.glob(src + "/*/*counter_collection.csv")
    kt = glob.glob(src + "/*/*kernel_trace.csv")
    if cc:
        rows = [r for r in csv.DictReader(open(cc[0])) if last_op(r["Kernel_Name"])]
        if rows:
            mx = max(int(r["Grid_Size"]) for r in rows)
            acc = collections.defaultdict(list)
            for r in rows:
                if int(r["Grid_Size"]) == mx:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["grid_threads"] = mx
            out["kernel"] = rows[0]["Kernel_Name"].split("(")[0]
            # two gathers of one group can share the grid bound (the feature buffer caps both: hop 2 and hop 3 at [15,10,5]);
            # the last hop's launches are the ones with the large counts: keep the launches within 25 % of the largest
            out["counters"] = {k: (lambda big: sum(big) / len(big))([x for x in v if x >= 0.75 * max(v)]) for k, v in acc.items()}
            out["launches"] = max(len([x for x in v if x >= 0.75 * max(v)]) for v in acc.values())
    if kt:
        rows = [r for r in csv.DictReader(open(kt[0])) if last_op(r["Kernel_Name"])]
        if rows:
            gx = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) for r in rows)
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows
                 if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == gx]
            d = sorted(x for x in d if x >= 0.75 * max(d))        # (same rule: the last hop's launches only)
            out["kernel_us_median"] = d[len(d) // 2]
            out["kernel_us_mean"] = sum(d) / len(d)
    try:
        b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
        out["bench"] = {"workload": b["config"]["workload"], "batches_per_launch_group": b["config"]["batches_per_launch_group"],
                        "rows_per_launch": b["roofline"]["rows_per_launch"], "bytes_per_row": b["roofline"]["bytes_per_row"],
                        "frac_hip_events": b["roofline"]["frac"], "avg_launch_us_hip_events": b["roofline"]["avg_launch_us"],
                        "value": b["value"], "feature_cache_hit_rate": b.get("feature_cache_hit_rate"),
                        "feature_cache_rows": b["config"].get("feature_cache_rows")}
    except Exception as e:
        out["bench_error"] = repr(e)[:200]
        try:
            out["stderr_tail"] = open(src + "/err.txt").read()[-600:]
        except OSError:
            pass
    json.dump(out, open(dst + "/fold.json", "w"), indent=1)


def table(root):
    print("| shape | rows / launch | algorithmic GB | read GB (request sizes) | read GB (2 x FETCH_SIZE) | written GB | traffic / algorithmic | "
          "kernel us (serialised, PMC pass) | algorithmic TB/s (frac of 8) | traffic TB/s | L2 hit | 32/64/128-B read requests (M) | "
          "UTCL1 miss / request | UTCL2 busy | wait_inst / active / wait_any (of wave cycles) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for shape in sorted(os.listdir(root)):
        folds = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(root, shape, "g*", "fold.json")))]
        if not folds:
            continue
        c, us, bench = {}, [], None
        for f in folds:
            c.update(f.get("counters", {}))
            if "kernel_us_median" in f:
                us.append(f["kernel_us_median"])
            bench = bench or f.get("bench")
        if not bench:
            print(f"| {shape} | no bench line | | | | | | | | | | | | | |")
            continue
        rows = bench["rows_per_launch"]
        alg = rows * bench["bytes_per_row"]
        rd_f = 2 * c.get("FETCH_SIZE", float("nan")) * 1024
        wr = c.get("WRITE_SIZE", float("nan")) * 1024
        r32, r64, r128 = (c.get("TCC_EA0_RDREQ_%s_sum" % s, float("nan")) for s in ("32B", "64B", "128B"))
        rd_all = c.get("TCC_EA0_RDREQ_sum", float("nan"))
        rd_q = r32 * 32 + r128 * 128 + (rd_all - r32 - r128) * 64       # everything that is neither 32 B nor 128 B counted at 64 B
        rd = rd_q if rd_q == rd_q else rd_f
        t = sorted(us)[len(us) // 2] if us else float("nan")
        hit = c.get("TCC_HIT_sum", float("nan")) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
        ut = c.get("TCP_UTCL1_TRANSLATION_MISS_sum", float("nan")) / max(c.get("TCP_UTCL1_REQUEST_sum", 1), 1)
        u2 = c.get("GRBM_UTCL2_BUSY", float("nan")) / max(c.get("GRBM_GUI_ACTIVE", 1), 1)
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
        sq = "%.2f / %.2f / %.2f" % (c.get("SQ_WAIT_INST_ANY", float("nan")) / wc, c.get("SQ_ACTIVE_INST_ANY", float("nan")) / wc,
                                     c.get("SQ_WAIT_ANY", float("nan")) / wc)
        print(f"| {shape} | {rows:,.0f} | {alg / 1e9:.3f} | {rd_q / 1e9:.3f} | {rd_f / 1e9:.3f} | {wr / 1e9:.3f} | {(rd + wr) / alg:.3f} | "
              f"{t:.0f} | {alg / t / 1e6:.2f} ({alg / t / 1e6 / 8:.3f}) | {(rd + wr) / t / 1e6:.2f} | {hit:.3f} | "
              f"{r32 / 1e6:.2f} / {r64 / 1e6:.2f} / {r128 / 1e6:.2f} of {rd_all / 1e6:.2f} | {ut:.4f} | {u2:.3f} | {sq} |")
        print(f"|  | {bench['workload']}; {bench['batches_per_launch_group']} batches per launch; HIP-event frac in that run "
              f"{bench['frac_hip_events']:.3f}; feature-cache hit rate {bench.get('feature_cache_hit_rate')} | | | | | | | | | | | | | |")


if __name__ == "__main__":
    if sys.argv[1] == "fold":
        fold(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        table(sys.argv[2])
