"""Build / load the CPU-emulation build of the kernel sources (TEST INFRASTRUCTURE ONLY, see tests/emu/)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(ROOT, "build", "emu", "libsegmamba_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def emu_available() -> bool:
    return os.path.exists(CLANG)


def build_emu(force: bool = False) -> str:
    srcs = glob.glob(os.path.join(ROOT, "segmamba_amd", "csrc", "*")) + glob.glob(os.path.join(EMU_DIR, "*.cpp")) \
        + glob.glob(os.path.join(EMU_DIR, "hip", "*.h")) + [os.path.join(ROOT, "include", "segmamba_hip.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flags = ["-O1", "-std=c++17", "-fPIC", "-pthread", "-I" + EMU_DIR, "-Wno-unused-value", "-DSEGM_PIN_F32(x)=",
             "-DSEGM_SCHED_FENCE()=", "-DSEGM_EMU=1", "-DSEGM_PIN_F2(x)=", "-DSEGM_WAVE_LDS_SYNC()=hipemu::sync_wave()",
             "-DSEGM_BLOCK_LDS_SYNC()=hipemu::sync_block()"]
    # one object per kernel translation unit (as in the product build), compiled in parallel as plain C++
    units = sorted(glob.glob(os.path.join(ROOT, "segmamba_amd", "csrc", "*.hip"))) + [os.path.join(EMU_DIR, "hip_emu_runtime.cpp")]
    objdir = os.path.join(os.path.dirname(OUT), "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        subprocess.run([CLANG, *flags, "-x", "c++", "-c", src, "-o", obj], check=True, cwd=ROOT)
        return obj
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, units))
    subprocess.run([CLANG, "-shared", "-pthread", *objs, "-o", OUT], check=True, cwd=ROOT)
    return OUT


_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        from segmamba_amd.lib import SegmLib
        _lib = SegmLib(build_emu())
    return _lib
