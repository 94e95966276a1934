#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average /
percentage (= what `--stats` reports), optionally split by grid size.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof1/r01_results.db [--by-grid] [--loop] > profiles/xxx.txt
--loop: only the dispatches after the last weight-packing kernel (pack_* / *_pack_*): the sampling loop without the
model set-up, so that busy / span says how much of the timed pass the GPU was executing kernels."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    c = db.cursor()
    key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
    where = ""
    if "--loop" in sys.argv:
        t0 = list(c.execute("select max(end) from kernels where name like '%pack%' or name like '%bn_affine%' or name like '%bn_fold%'"))[0][0]
        if t0 is not None:
            where = f"where start > {t0}"
    rows = list(c.execute(f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          f"from kernels {where} group by {key} order by sum(duration) desc"))
    total = sum(r[-4] for r in rows)
    span = list(c.execute(f"select min(start), max(end) from kernels {where}"))[0]
    print(f"# kernels: {sum(r[-5] for r in rows)} dispatches, busy {total / 1e6:.1f} ms, span {(span[1] - span[0]) / 1e6:.1f} ms")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  name")
    for r in rows[:80]:
        if by_grid:
            name = f"{r[0][:90]} grid=({r[1]},{r[2]},{r[3]})"
            n, tot, avg, mn, mx = r[4:]
        else:
            name = r[0][:110]
            n, tot, avg, mn, mx = r[1:]
        print(f"{n:8d} {tot / 1e6:10.2f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}  {name}")


if __name__ == "__main__":
    main()
