#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average /
percentage (= what `--stats` reports), optionally split by grid size.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof1/r01_results.db [--by-grid] [--loop] > profiles/xxx.txt
--loop: only the dispatches after the last weight-packing kernel (pack_* / *_pack_*): the sampling loop without the
model set-up, so that busy / span says how much of the timed pass the GPU was executing kernels.
--gaps: idle time between consecutive dispatches (end of one -> start of the next, overlaps count as zero): histogram,
the idle time attributed to the kernel that FOLLOWS the gap (who was late), and the largest individual gaps."""
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
    if "--gaps" in sys.argv:
        return gaps(c, where)
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


def gaps(c, where):
    ks = list(c.execute(f"select start, end, name from kernels {where} order by start"))
    edges = [1, 5, 10, 20, 50, 100, 500, 1000, 10000, 100000, 10 ** 9]
    hist = [[0, 0.0] for _ in edges]
    late = {}
    big = []
    busy_end = ks[0][1]
    idle = 0.0
    for (s, e, name) in ks[1:]:
        g = max(0, s - busy_end) / 1e3          # us
        busy_end = max(busy_end, e)
        idle += g
        for i, lim in enumerate(edges):
            if g < lim:
                hist[i][0] += 1
                hist[i][1] += g
                break
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        a = late.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += g
        if g >= 200:
            big.append((g, short))
    span = (ks[-1][1] - ks[0][0]) / 1e3
    print(f"# {len(ks)} dispatches, span {span / 1e3:.1f} ms, idle between dispatches {idle / 1e3:.1f} ms ({100 * idle / span:.1f} %)")
    print("# gap histogram: upper bound (us), count, total idle ms")
    for lim, (n, t) in zip(edges, hist):
        print(f"  <{lim:>10}  {n:8d}  {t / 1e3:10.2f}")
    print("# idle time by the kernel that follows the gap (top 25): count, idle ms, mean gap us")
    for k, (n, t) in sorted(late.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {n:8d} {t / 1e3:10.2f} {t / n:9.1f}  {k}")
    print(f"# gaps >= 200 us: {len(big)}, total {sum(g for g, _ in big) / 1e3:.1f} ms; the 20 largest:")
    for g, k in sorted(big, reverse=True)[:20]:
        print(f"  {g / 1e3:10.2f} ms before {k}")


if __name__ == "__main__":
    main()
