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
        print("| kernel | calls | total_us | avg_us | min_us | max_us | pct | vgpr | lds | wg |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        rows = db.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, max(vgpr_count), max(lds_size), max(workgroup_x) "
            "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows)
        for r in rows:
            print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f | %d | %d | %d |" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8]))
    else:
        print("| kernel | counter | dispatches | avg | min | max |")
        print("|---|---|---|---|---|---|")
        for r in db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                            "group by kernel_name, counter_name order by avg(value) desc"):
            print("| %s | %s | %d | %.1f | %.1f | %.1f |" % (short(r[0]), r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
