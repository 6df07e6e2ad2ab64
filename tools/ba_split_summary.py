#!/usr/bin/env python
"""Condenses the raw DMVIO_HIP_BA_TIMING=1 log of a bench run ([dmvio_hip_ba] lines: one block per handle that was destroyed) and the kernel-timeline slice of
tools/rocprof_timeline.py into a summary of at most 60 lines: per phase the median / min / max over the handles' blocks, weighted by nothing — every block is a mean over
its own iterations already —, then the first two accepted iterations of the timeline.
usage: ba_split_summary.py <title> <timing.log> [timeline.txt] [ba_loop.log]"""
import re
import sys
import numpy as np


def parse_pairs(s):
    out = []
    for m in re.finditer(r"([A-Za-z+/ :\-]+?)=([0-9.]+)", s):
        k = m.group(1).strip()
        if k and k != "-":
            out.append((k, float(m.group(2))))
    return out


def table(title, blocks, min_n=1):
    if not blocks:
        return []
    keys = [k for k, _ in blocks[0][1]]
    rows = ["", "## " + title, "", "| phase | median us | min | max |", "|---|---|---|---|"]
    for k in keys:
        v = np.array([dict(b[1]).get(k, np.nan) for b in blocks if b[0] >= min_n])
        v = v[~np.isnan(v)]
        if len(v):
            rows.append("| %s | %.1f | %.1f | %.1f |" % (k, np.median(v), v.min(), v.max()))
    return rows


def main():
    title, log = sys.argv[1], sys.argv[2]
    gn, sg, acc = [], [], {}
    for line in open(log, errors="replace"):
        if "[dmvio_hip_ba]" not in line:
            continue
        m = re.search(r"GN iteration.*over (\d+) iterations \(us/iter\): (.*)", line)
        if m:
            gn.append((int(m.group(1)), parse_pairs(m.group(2)))); continue
        m = re.search(r"set_graph over (\d+) calls \(us/call\): (.*)", line)
        if m:
            sg.append((int(m.group(1)), parse_pairs(m.group(2)))); continue
        m = re.search(r"k_ba_accumulate (.*?) blocks=(\d+): duration mean ([0-9.]+) max ([0-9.]+) us .*latest end ([0-9.]+) us", line)
        if m:
            acc.setdefault("%s (%s workgroups)" % (m.group(1), m.group(2)), []).append((float(m.group(3)), float(m.group(4)), float(m.group(5))))
    out = ["# " + title, "",
           "Condensed by tools/ba_split_summary.py from %d per-handle blocks of the raw `DMVIO_HIP_BA_TIMING=1` log of one `bench.py` run (each block: means over the handle's own" % len(gn),
           "iterations, host clock between phases, no synchronisation added; the 'wait' and 'accepted / rejected' phases include the kernels they wait for)."]
    fresh = [b for b in gn if b[0] <= 60]
    loop = [b for b in gn if b[0] > 60]
    out += table("GN iteration on fresh windows (handles with <= 60 iterations: optimize(6), every step accepted)", fresh)
    out += table("GN iteration in the converged loop (handles with > 60 iterations: steps mostly rejected)", loop)
    if sg:
        keys = [k for k, _ in sg[0][1]]
        out += ["", "dmvio_hip_ba_set_graph, median us per call: " + ", ".join("%s %.1f" % (k, np.median([dict(b[1]).get(k, np.nan) for b in sg])) for k in keys)]
    if acc:
        out += ["", "k_ba_accumulate, in-kernel stamps per role, medians (mean duration / longest workgroup / last workgroup ends at, us): " +
                "; ".join("%s %.1f / %.1f / %.1f" % ((k,) + tuple(np.median(np.array(v), axis=0))) for k, v in acc.items())]
    if len(sys.argv) > 4:
        keep = [l.rstrip() for l in open(sys.argv[4], errors="replace") if re.search(r"GN-iterations/s|optimize\(6\) on the fresh|decision pass", l)]
        if keep:
            out += ["", "## tools/ba_loop.py under rocprofv3 --kernel-trace (the profiler adds ~15 %)", ""] + keep[:4]
    if len(sys.argv) > 3:
        tl = [l.rstrip() for l in open(sys.argv[3], errors="replace") if " dur " in l]
        if tl:
            out += ["", "## kernel timeline of two accepted iterations (us): start, duration, gap to the previous kernel, workgroups", ""] + tl[:14]
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
