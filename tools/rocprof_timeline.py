#!/usr/bin/env python
"""Prints a slice of the kernel timeline of a rocprofv3 rocpd database: start offset, duration and the gap to the previous kernel (us).
usage: rocprof_timeline.py <results.db> [first_row | kernel-name substring] [n_rows] [occurrence]
With a kernel name the slice starts two rows before the `occurrence`-th launch (default 3) of a kernel whose name contains it."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sel = sys.argv[2] if len(sys.argv) > 2 else "2000"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
occ = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = db.execute("select name, start, end, grid_x / workgroup_x from kernels order by start").fetchall()
if sel.isdigit():
    first = int(sel)
else:
    hits = [i for i, r in enumerate(rows) if sel in r[0]]
    if not hits:
        sys.exit("no kernel named *%s*" % sel)
    first = max(0, hits[min(occ, len(hits)) - 1] - 2)
rows = rows[first:first + n]
t0 = rows[0][1]; prev_end = rows[0][1]
for name, s, e, wg in rows:
    nm = name.replace("void ", "").split("(")[0]
    print("%9.2f  dur %7.2f  gap %7.2f  wg %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, wg, nm))
    prev_end = e
