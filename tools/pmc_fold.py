"""Folds rocprofv3 --pmc passes (one counter group per run, `--kernel-trace` alongside) into ONE table per kernel:
    python tools/pmc_fold.py <dir with g1/, g2/, ... pass directories> <out.md> [kernel-name substrings ...]
Every pass directory holds a `*counter_collection.csv` (and `*kernel_trace.csv`) of the same command.  Kernels are keyed by
(name, grid); for each key the counters are averaged over the launches of the pass that collected them.  Derived columns use the
guide's gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE is doubled (128-byte requests
tallied at 64 bytes); TCC_* are summed over the channels."""
import collections
import csv
import glob
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    wanted = sys.argv[3:] or ["sample_kernel", "dedup_lds_kernel", "dedup_lists_kernel", "compact_kernel", "place_kernel", "list_known_kernel", "gather_kernel"]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))     # (kernel, grid) -> counter -> [sum, n]
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in sorted(glob.glob(os.path.join(src, "g*"))):
        for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if not any(w in n for w in wanted):
                    continue
                key = (n[:n.rfind("(")] if "(" in n else n, int(r["Grid_Size"]))
                a = acc[key][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                if "Start_Timestamp" in r and r["Counter_Name"] == "SQ_WAVES":      # (one row per dispatch: the pass that counts waves)
                    t = dur[key]
                    t[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; t[1] += 1
    def c(key, name):
        a = acc[key].get(name)
        return a[0] / a[1] if a and a[1] else None
    lines = ["| kernel | grid (threads) | launches | µs (under the counters) | HBM read MB | HBM written MB | TB/s | L2 hit rate | L2 requests M | "
             "memory-side atomics | waves | wait-any % of wave cycles | LDS wait % | LDS bank-conflict % of LDS cycles | VALU / VMEM / LDS insts per wave |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(acc, key=lambda k: (k[0], -k[1])):
        us = dur[key][0] / dur[key][1] if dur[key][1] else None
        fetch, write = c(key, "FETCH_SIZE"), c(key, "WRITE_SIZE")
        rd = 2 * fetch * 1024 / 1e6 if fetch is not None else None
        wr = write * 1024 / 1e6 if write is not None else None
        hit, miss, req, atom = c(key, "TCC_HIT_sum"), c(key, "TCC_MISS_sum"), c(key, "TCC_REQ_sum"), c(key, "TCC_ATOMIC_sum")
        waves, wcyc, wany = c(key, "SQ_WAVES"), c(key, "SQ_WAVE_CYCLES"), c(key, "SQ_WAIT_ANY")
        wlds, bank, ldsact = c(key, "SQ_WAIT_INST_LDS"), c(key, "SQ_LDS_BANK_CONFLICT"), c(key, "SQ_LDS_IDX_ACTIVE")
        valu, vmem, lds = c(key, "SQ_INSTS_VALU"), c(key, "SQ_INSTS_VMEM"), c(key, "SQ_INSTS_LDS")
        n_l = max((a[1] for a in acc[key].values()), default=0)
        f = lambda v, fmt="{:.1f}": "" if v is None else fmt.format(v)
        tbs = (rd + wr) / us if (rd is not None and wr is not None and us) else None     # MB / us = TB/s
        lines.append("| `{}` | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |".format(
            key[0].replace("void ", "").replace("lg::", ""), key[1], n_l, f(us), f(rd), f(wr), f(tbs, "{:.2f}"),
            f(hit / (hit + miss) if hit is not None and miss is not None and hit + miss > 0 else None, "{:.2f}"),
            f(req / 1e6 if req is not None else None, "{:.2f}"), f(atom, "{:.0f}"), f(waves, "{:.0f}"),
            f(100 * wany / wcyc if wany is not None and wcyc else None), f(100 * wlds / wcyc if wlds is not None and wcyc else None),
            f(100 * bank / ldsact if bank is not None and ldsact else None),
            "" if None in (valu, vmem, lds, waves) or not waves else f"{valu / waves:.0f} / {vmem / waves:.0f} / {lds / waves:.0f}"))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
