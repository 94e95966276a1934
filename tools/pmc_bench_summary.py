#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel-class HBM bytes
per launch (profiles/pmc_summary.json, read by bench.py for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte requests as
64 B, i.e. exactly half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section), so it
is doubled; WRITE_SIZE matched the algorithmic output size of the FF1 GEMM to <1 % in our
calibration launch (tools/pmc_probe.py) and is used as is."""
import collections
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash  # noqa: E402


def klass(name):
    if "pgemm_kernel" in name:          # the persistent linear kernel (csrc/pgemm.hip)
        return "linear_gemm"
    if "pconv_kernel" in name:          # the persistent row-sharing 3x3 kernel (csrc/pconv.hip)
        return "conv3x3_gemm"
    if "igemm_kernel" in name:
        args = [a.strip() for a in name.split("igemm_kernel<")[1].split(">")[0].split(",")]   # BM, BN, MODE
        return "linear_gemm" if args[2] == "0" else "conv3x3_gemm"
    if "ffn_chain_kernel" in name or "lin_chain_kernel" in name:      # the one-kernel token-local chains of the C = 320 level
        return "linear_gemm"
    if "splitk_reduce" in name:
        return "splitk_reduce"
    if "self_attn" in name:
        return "self_attn"
    if "cross_attn" in name:
        return "cross_attn"
    if "gn_" in name or "layernorm" in name:
        return "norm"
    if "geglu" in name or "concat" in name:
        return "other"
    return None


def read(dbpath, counter):
    db = sqlite3.connect(dbpath)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    ix = {k: i for i, k in enumerate(cols)}
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        k = klass(r[ix.get("kernel_name", ix.get("name"))])
        if k:
            agg[k][0] += r[ix["value"]]
            agg[k][1] += 1
    return agg


def bench_config(log):
    """config.pmc_config of the bench JSON line in the profiled run's stdout"""
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"pmc_config"' in line:
            return json.loads(line)["config"]["pmc_config"]
    return None


def main():
    fetch = read(sys.argv[1], "FETCH_SIZE")
    write = read(sys.argv[2], "WRITE_SIZE")
    cfg = bench_config(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        n = max(nf, nw, 1)
        out[k] = {"launches": n, "fetch_kib_raw_per_launch": f / max(nf, 1), "write_kib_per_launch": w / max(nw, 1),
                  "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    json.dump({"config": cfg, "csrc_sha256": csrc_hash(), "counters": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 counts 128-B requests as 64 B) + WRITE_SIZE, KiB",
               "classes": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
