#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
rows = list(c.execute("select * from counters_collection"))
ix = {k: i for i, k in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix.get('kernel_name', ix.get('name'))]
    want = sys.argv[2:] or ['igemm', 'self_attn', 'cross_attn', 'gn_', 'ffn_', 'lin_chain', 'layernorm']
    if not any(w in name for w in want):
        continue
    key = (name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60], r[ix['grid_size_x']] if 'grid_size_x' in ix else 0)
    agg[key][r[ix['counter_name']]].append(r[ix['value']])
for key, d in agg.items():
    print(key)
    for k, v in sorted(d.items()):
        print(f"    {k:32s} n={len(v):3d} avg={sum(v) / len(v):.4g}")
