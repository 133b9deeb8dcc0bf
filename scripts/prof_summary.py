#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (…_results.db) into a small text table:
per-kernel calls, total/avg/min/max duration (µs), share.  Usage:
    python scripts/prof_summary.py gpurun_out/<tag>/prof/bench_results.db [> profiles/<name>.txt]
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# {'kernel':60s} {'calls':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'share%':>7s} {'vgpr':>4s} {'sgpr':>4s} {'lds':>6s} {'scr':>4s} {'grid':>8s} {'wg':>4s}")
    for name, calls, tot, avg, mn, mx, vg, sg, lds, scr, gx, wx in rows:
        short = name.split("(")[0]
        if len(short) > 60:
            short = short[:57] + "..."
        print(f"  {short:60s} {calls:5d} {avg / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} {100 * tot / total:7.2f} "
              f"{vg or 0:4d} {sg or 0:4d} {lds or 0:6d} {scr or 0:4d} {gx or 0:8d} {wx or 0:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
