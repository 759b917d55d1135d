"""Folds a rocprofv3 `--kernel-trace [--marker-trace]` csv directory of `bench.py` (weave arrangement) into the timeline of ONE
steady-state launch group: which kernels run on the heavy stream, in which order and for how long, what the light stream's
kernels (the next group's head) add up to and how much of that is hidden under heavy kernels, and where the heavy stream idles.

    cd /tmp && rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 8 --warmup 2 ...
    python tools/trace_group.py /tmp/tl [out.md]

A group's interval = from the end of one LASTOP gather (`gather_kernel<..., true>`, the last kernel of a group's REST phase) to the
end of the next; medians over the groups of the second half of the trace (the timed replays).  With `--marker-trace` the host-side
`group slot=...` ranges (markers.hip) are counted and their lengths reported: what a group's submission costs the host."""
import collections
import csv
import glob
import statistics
import sys


def short(n):
    n = n[:n.rfind("(")] if "(" in n else n
    return n.replace("void ", "").replace("lg::", "").replace("float __vector(4)", "float4").replace(" ", "")


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    ks = []
    for f in glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "lg::" not in r["Kernel_Name"]:
                continue
            gy = r.get("Grid_Size_Y", "1")
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "")),
                       int(gy) if gy.isdigit() else 1))
    ks.sort()
    lanes = max(k[4] for k in ks)
    ks = [k for k in ks if k[4] == lanes]                       # launches over a full group
    last = [k for k in ks if "gather_kernel" in k[2] and k[2].endswith("true>")]
    heavy_q = collections.Counter(k[3] for k in last).most_common(1)[0][0]
    last = [k for k in last if k[3] == heavy_q]
    last = last[len(last) // 2:]
    # the light stream = the queue the groups' heads run on (it starts with batch_generate_kernel); every other queue belongs to the REST
    # phase: the heavy stream and, with three hops and more, the branch of its graph that runs the last hop's de-duplication beside the
    # earlier hops' gathers
    heads = collections.Counter(k[3] for k in ks if "batch_generate_kernel" in k[2])
    light_q = heads.most_common(1)[0][0] if heads else None
    if light_q == heavy_q:
        light_q = None
    groups = []
    for a, b in zip(last, last[1:]):
        t0, t1 = a[1], b[1]
        heavy = [k for k in ks if k[3] != light_q and t0 <= k[0] < t1]
        light = [k for k in ks if k[3] == light_q and k[0] < t1 and k[1] > t0]
        # time of the interval in which at least one kernel of the REST phase runs (union: its branches overlap)
        busy, reach = 0, t0
        for k in heavy:
            lo, hi = max(k[0], reach), min(k[1], t1)
            if hi > lo:
                busy += hi - lo
            reach = max(reach, hi)
        light_busy = sum(min(k[1], t1) - max(k[0], t0) for k in light)
        hidden = 0
        for l in light:
            reach = t0
            for h in heavy:
                lo, hi = max(l[0], h[0], t0, reach), min(l[1], h[1], t1)
                if hi > lo:
                    hidden += hi - lo
                reach = max(reach, min(h[1], t1))
        seq = []
        reach = t0
        for k in heavy:
            seq.append((k[2] + ("" if k[3] == heavy_q else " (side branch)"), (k[1] - k[0]) / 1e3, (k[0] - reach) / 1e3))
            reach = max(reach, k[1])
        groups.append({"step": (t1 - t0) / 1e3, "busy": busy / 1e3, "light": light_busy / 1e3, "hidden": hidden / 1e3, "seq": seq,
                       "light_names": [k[2] for k in light]})
    if not groups:
        print("no steady-state groups found")
        return
    n_seq = collections.Counter(len(g["seq"]) for g in groups).most_common(1)[0][0]
    groups = [g for g in groups if len(g["seq"]) == n_seq]
    med = lambda xs: statistics.median(xs)
    lines = [f"Launch group of {lanes} lanes, heavy stream = queue {heavy_q}; medians over {len(groups)} replayed groups.", "",
             "| # | kernel of the REST phase, by start | µs | idle before it, µs (negative: it starts while an earlier one still runs) |", "|---|---|---|---|"]
    for i in range(n_seq):
        lines.append("| {} | `{}` | {:.1f} | {:.1f} |".format(i, groups[0]["seq"][i][0], med([g["seq"][i][1] for g in groups]),
                                                              med([g["seq"][i][2] for g in groups])))
    step, busy = med([g["step"] for g in groups]), med([g["busy"] for g in groups])
    light, hidden = med([g["light"] for g in groups]), med([g["hidden"] for g in groups])
    gathers = med([sum(s[1] for s in g["seq"] if "gather_kernel" in s[0]) for g in groups])
    sampler = med([sum(s[1] for s in g["seq"] if "gather_kernel" not in s[0]) for g in groups])
    lines += ["", f"step {step:.1f} µs = REST phase busy {busy:.1f} (its kernels add up to: gathers {gathers:.1f} + sampler chain {sampler:.1f}"
                  f"{'' if abs(gathers + sampler - busy) < 1 else ', overlapping by %.1f' % (gathers + sampler - busy)}) + idle {step - busy:.1f}; "
                  f"light stream (the next group's head: {len(set(groups[0]['light_names']))} kernel kinds) busy {light:.1f} µs, "
                  f"{hidden:.1f} of them under a heavy kernel."]
    marks = []
    for f in glob.glob(d + "/*/*marker_api_trace.csv") + glob.glob(d + "/*marker_api_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "group slot=" in r.get("Function", ""):
                marks.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if marks:
        lines += ["", f"{len(marks)} `group` ranges (roctx, host side): submitting one group takes {statistics.median(marks):.1f} µs of host time (median)."]
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main()
