#!/usr/bin/env python3
"""Golden vectors for the PIE-Bench evaluator (build container only: imports the reference from /root/reference).
Runs the reference's own ``mask_decode`` (text-guided/evaluation/evaluation.py:9-25, unmodified; the module's top-level
import of its torchmetrics-based calculator is the only line not executed) on seeded run-length masks and stores
inputs + outputs in tests/golden/g15_mask_decode.npz.     python tests/golden/make_golden_eval.py"""
import os

import numpy as np

REF = "/root/reference/text-guided/evaluation/evaluation.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g15_mask_decode.npz")


def main():
    src = open(REF).read().split("def calculate_metric")[0].replace("from matrics_calculator import MetricsCalculator", "")
    ns = {}
    exec(compile(src, REF, "exec"), ns)
    rng = np.random.default_rng(0)
    out = {}
    for i in range(6):
        starts = np.sort(rng.integers(0, 512 * 512, size=5 + i))
        enc = []
        for s in starts:
            enc += [int(s), int(rng.integers(1, 5000))]
        if i == 5:
            enc += [512 * 512 - 10, 400]            # a run that overshoots the image
        out[f"enc{i}"] = np.asarray(enc, dtype=np.int64)
        out[f"mask{i}"] = np.packbits(ns["mask_decode"](enc).astype(np.uint8))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
