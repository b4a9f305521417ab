"""Build the C restatement of the oracle (oracle/scan_ref.c) into oracle/_build/libscan_ref.so with gcc + OpenMP.

    python oracle/build_oracle.py [--force]

Test infrastructure: the shared object is only ever loaded by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (through oracle/scan_ref.py); building the checker is not using it.  The reference itself is Python
(its native CUDA kernels cannot be compiled here), so there is no oracle/_ref build.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "scan_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libscan_ref.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-Wall", SRC, "-o", OUT, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed for the C oracle:\n" + r.stderr[-4000:])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
