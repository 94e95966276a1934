"""Compare the gfx950 ISA of csrc/ units between a git revision and the working tree, kernel by kernel (labels and mangled
names normalised): the check behind "this refactor does not touch the bfloat16 kernels".
`python tools/isa_diff.py [--rev HEAD] [--define X] unit ...`  (unit = gemm, ffn, ...)"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "h-edit_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "--cuda-device-only", "-S"]
UNIT_FLAGS = {"attn": ["-fno-honor-nans"], "ffn": ["-fno-honor-nans"]}


def kernels(path):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[cur] = buf
            cur = None
            continue
        l = line.split(";")[0].strip()
        if l and not l.startswith("."):
            buf.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r"_Z\w+", "SYM", l)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rev", default="HEAD")
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("units", nargs="+")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="isa_diff_")
    old = os.path.join(tmp, "old")
    os.makedirs(os.path.join(old, "h-edit_amd", "csrc"))
    os.makedirs(os.path.join(old, "include"))
    files = subprocess.check_output(["git", "ls-tree", "-r", "--name-only", a.rev, "h-edit_amd/csrc", "include"], cwd=ROOT, text=True).split()
    for f in files:
        os.makedirs(os.path.dirname(os.path.join(old, f)), exist_ok=True)
        open(os.path.join(old, f), "wb").write(subprocess.check_output(["git", "show", f"{a.rev}:{f}"], cwd=ROOT))
    defs = ["-D" + d for d in a.define]

    def cc(job):
        src, out, extra = job
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-3000:])
        return out
    jobs = []
    for u in a.units:
        jobs.append((os.path.join(old, "h-edit_amd", "csrc", u + ".hip"), os.path.join(tmp, u + "_old.s"), UNIT_FLAGS.get(u, [])))
        jobs.append((os.path.join(SRC, u + ".hip"), os.path.join(tmp, u + "_new.s"), UNIT_FLAGS.get(u, []) + defs))
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(cc, jobs))
    bad = 0
    for u in a.units:
        ko, kn = kernels(os.path.join(tmp, u + "_old.s")), kernels(os.path.join(tmp, u + "_new.s"))
        # a template parameter appended with its default: ...Lb0EE -> ...Lb0ELb0EE; match by prefix
        same = diff = missing = 0
        for name, body in ko.items():
            strip = lambda n: re.sub(r"(Lb0E)+(?=EEv)", "", n)           # trailing `false` template arguments
            cand = [n for n in kn if n == name] or [n for n in kn if strip(n) == strip(name)]
            if not cand:
                missing += 1
                print(f"{u}: {name} is gone")
                continue
            if any(kn[c] == body for c in cand):
                same += 1
            else:
                diff += 1
                print(f"{u}: {name} DIFFERS ({len(body)} -> {len(kn[cand[0]])} instructions)")
        print(f"{u}: {same} kernels identical, {diff} differ, {missing} missing, {len(kn) - len(ko)} new")
        bad += diff + missing
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
