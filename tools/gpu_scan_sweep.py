"""Forward / backward scan timing at the roofline shape for library variants / settings."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
from bench import scan_roofline
for dt in (torch.bfloat16, torch.float32):
    r = scan_roofline(dt, torch.device("cuda"))
    print("   ", dt, "chunk", r["shape"]["chunk"], "fwd ms", r["ms"], "frac", r["frac"], "bwd ms", r["backward"]["ms"], flush=True)
''' % ROOT
V = os.path.join(ROOT, "build", "variants")
for env in ({"SEGM_SCAN_FAST": "0"}, {}, {"SEGM_LIB_OUT": V + "/libsegm_w3.so"}, {"SEGM_LIB_OUT": V + "/libsegm_w4.so"},
            {"SEGM_CHUNK": "128"}, {"SEGM_CHUNK": "512"}):
    print("env", env, flush=True)
    subprocess.run([sys.executable, "-c", code], env={**os.environ, **env})
