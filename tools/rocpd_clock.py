#!/usr/bin/env python3
"""Shader clock per kernel from a rocprofv3 run with --kernel-trace --pmc SQ_BUSY_CYCLES:
GHz = SQ_BUSY_CYCLES / 32 shader engines / kernel duration.  usage: rocpd_clock.py results.db [name-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
ix = {k: i for i, k in enumerate(cols)}
rows = list(c.execute("select * from counters_collection"))
agg = collections.defaultdict(list)
for r in rows:
    if r[ix['counter_name']] != 'SQ_BUSY_CYCLES':
        continue
    name = r[ix.get('kernel_name', ix.get('name'))]
    if pat not in name:
        continue
    dur = r[ix['end']] - r[ix['start']] if 'end' in ix and 'start' in ix else None
    agg[(name.split('::')[-1][:44], r[ix.get('grid_size_x', 0)])].append((r[ix['value']], dur))
for k, v in agg.items():
    busy = sum(a for a, _ in v) / len(v)
    durs = [d for _, d in v if d]
    if durs:
        d = sum(durs) / len(durs)
        print(f"{k[0]:46s} grid {k[1]:>9} n={len(v):3d}  {d / 1e3:9.1f} us  busy/32 = {busy / 32:10.0f} cyc  -> {busy / 32 / d:5.2f} GHz")
    else:
        print(k, "no duration columns:", cols)
