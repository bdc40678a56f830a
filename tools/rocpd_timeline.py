#!/usr/bin/env python3
"""Kernel timeline of the LAST `n` steps in a rocprofv3 --kernel-trace rocpd database: every kernel with its stream, start and
duration relative to the first k_trace shown (what overlaps what, where the streams wait for each other).
Usage: tools/rocpd_timeline.py <results.db> [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tr = [r[0] for r in c.execute("select start from kernels where name like '%k_trace%' order by start desc limit ?", (n + 1,))]
t0 = tr[-1]
t1 = tr[0]
rows = list(c.execute("select name, stream_id, start, end from kernels where start >= ? and start < ? order by start", (t0, t1)))
streams = sorted({r[1] for r in rows})
for name, sid, s, e in rows:
    short = name.replace("void ", "").split("(")[0].split("<")[0]
    col = streams.index(sid)
    print("%9.1f  %7.1f us  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, "                    " * col, short))
print("streams (columns): %s; span of %d steps: %.1f us = %.1f us per step" % (streams, n, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / n))
