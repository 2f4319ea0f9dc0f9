#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite
output, ROCm 7.2 default) as a plain-text table for profiles/.   usage: rocpd_summary.py X_results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    lines = ["%-100s %7s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
         "group by name order by sum(duration) desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        lines.append("%-100s %7d %14d %12.0f %12d %12d %7.2f" % (name[:100], n, s, a, mn, mx, 100.0 * s / tot))
    # per-kernel launch geometry / registers of the first dispatch
    lines.append("")
    lines.append("%-60s %10s %8s %6s %6s %8s" % ("kernel", "grid_x", "wg_x", "vgpr", "sgpr", "lds"))
    seen = set()
    for r in cur.execute("select name, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels order by start"):
        if r[0] in seen:
            continue
        seen.add(r[0])
        lines.append("%-60s %10d %8d %6d %6d %8d" % (r[0][:60], r[1], r[2], r[3], r[4], r[5]))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
