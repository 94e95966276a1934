#!/usr/bin/env python3
"""sha256 over the kernel sources (h-edit_amd/csrc/*.hip, *.h and include/hedit.h; names and contents, sorted): the stamp that ties a
measurement file (profiles/pmc_summary.json) to the code it was taken on.  bench.py refuses a summary whose stamp is not the tree's."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    src = os.path.join(root, "h-edit_amd", "csrc")
    files = sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".h")))
    files.append(os.path.join(root, "include", "hedit.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_hash(sys.argv[1] if len(sys.argv) > 1 else ROOT))
