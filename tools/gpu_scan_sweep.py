"""Forward / backward scan timing at the roofline shape for a few (chunk, apply sub-tile) settings."""
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
for env in ({}, {"SEGM_CHUNK": "384"}, {"SEGM_CHUNK": "512"}, {"SEGM_CHUNK": "128"}, {"SEGM_APPLY_TS": "4"},
            {"SEGM_CHUNK": "384", "SEGM_APPLY_TS": "4"}):
    print("env", env, flush=True)
    subprocess.run([sys.executable, "-c", code], env={**os.environ, **env})
