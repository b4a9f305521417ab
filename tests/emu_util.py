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
    cmd = [CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + EMU_DIR, "-Wno-unused-value", "-DSEGM_PIN_F32(x)=", "-DSEGM_SCHED_FENCE()=", "-DSEGM_EMU=1", "-DSEGM_PIN_F2(x)=", "-DSEGM_WAVE_LDS_SYNC()=hipemu::sync_wave()",
           os.path.join(EMU_DIR, "emu_entry.cpp"), os.path.join(EMU_DIR, "hip_emu_runtime.cpp"), "-o", OUT]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return OUT


_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        from segmamba_amd.lib import SegmLib
        _lib = SegmLib(build_emu())
    return _lib
