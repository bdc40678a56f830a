#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace rocpd database (<dir>/<name>_results.db): per-kernel statistics and, for the RCCL
loopback profile, the order of the kernels on ONE handle's stream around the first RCCL kernels (RCCL's ncclDevKernel_*
must sit between k_pack / k_shard_publish and k_unpack_*, and between the fusion and k_map2d).
Usage: tools/rocpd_summary.py <results.db> [--stream-excerpt N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    n_ex = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--stream-excerpt" else 0
    print("%-72s %7s %10s %9s %9s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
    rows = list(c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc"))
    tot = float(sum(r[5] for r in rows)) or 1.0
    for r in rows:
        print("%-72s %7d %10.2f %9.2f %9.2f %6.1f%%" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot))
    if n_ex:
        hit = c.execute("select stream_id, start from kernels where name like 'ncclDevKernel%' order by start limit 1").fetchone()
        if not hit:
            print("\n(no RCCL kernel in this trace)")
            return
        sid, t0 = hit
        print("\nstream %s around its first RCCL kernels (start relative to the first one, us):" % sid)
        before = list(c.execute("select name, start, end from kernels where stream_id=? and start<? order by start desc limit 4", (sid, t0)))[::-1]
        after = list(c.execute("select name, start, end from kernels where stream_id=? and start>=? order by start limit ?", (sid, t0, n_ex)))
        for name, s, e in before + after:
            print("  %+10.1f  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name[:90]))
        streams = c.execute("select count(distinct stream_id) from kernels where name like 'ncclDevKernel%'").fetchone()[0]
        print("RCCL kernels ran on %d distinct streams (one per rank handle)" % streams)


if __name__ == "__main__":
    main()
