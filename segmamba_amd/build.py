"""Build libsegmamba_hip.so in-tree with hipcc for gfx950 (MI355X).

    python -m segmamba_amd.build [--force]

One object per .hip translation unit (compiled in parallel), linked into
segmamba_amd/libsegmamba_hip.so next to this file so that it travels with the source tree
(git-ignored, not gpurun-ignored).  No GPU is needed to build: hipcc cross-compiles.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_PATH = os.path.join(HERE, "libsegmamba_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = os.environ.get("SEGM_EXTRA_HIPCC_FLAGS", "").split()      # experiments only (e.g. -DSEGM_BWD_MIN_WAVES=2)
if os.environ.get("SEGM_LIB_OUT"):
    LIB_PATH = os.environ["SEGM_LIB_OUT"]
    OBJ_DIR = OBJ_DIR + "_" + os.path.basename(LIB_PATH)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "segmamba_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def _compile(src: str) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".o"))
    hdr_t = max(os.path.getmtime(p) for p in _deps() if not p.endswith(".hip"))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
        return obj
    cmd = [HIPCC, *FLAGS, *EXTRA, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"{HIPCC} not found: cannot build the HIP library (and there is no fallback path)")
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(o)
    srcs = _sources()
    if verbose:
        print(f"[segmamba_amd.build] hipcc {ARCH}: {len(srcs)} translation units", flush=True)
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[segmamba_amd.build] wrote {LIB_PATH}", flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
