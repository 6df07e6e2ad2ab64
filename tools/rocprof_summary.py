#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (sqlite) output into small CSV/markdown files that can be committed under profiles/.

usage: rocprof_summary.py <results.db> [--counters] > summary.md
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    counters = "--counters" in sys.argv
    if not counters:
        # one row per (kernel, grid): the same kernel launched over different problem sizes (batch 1024 vs the 64-frame overlap leg) stays apart
        print("| kernel | workgroups | calls | total_us | avg_us | min_us | max_us | pct | vgpr | lds | wg |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        rows = db.execute(
            "select name, (grid_x / workgroup_x) * max(grid_y / workgroup_y, 1) as wgs, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
            "max(vgpr_count), max(lds_size), max(workgroup_x) from kernels group by name, wgs order by sum(duration) desc").fetchall()
        tot = sum(r[3] for r in rows)
        for r in rows[:40]:
            print("| %s | %d | %d | %.1f | %.2f | %.2f | %.2f | %.2f | %d | %d | %d |" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], r[6], 100 * r[3] / tot, r[7], r[8], r[9]))
    else:
        print("| kernel | counter | dispatches | avg | min | max |")
        print("|---|---|---|---|---|---|")
        for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                            "group by kernel_name, counter_name order by avg(value) desc"):
            print("| %s | %s | %d | %.1f | %.1f | %.1f |" % (short(r[0]), r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
