"""A/B timing of library variants of the channel-last 3x3x3 forward at 48 -> 48 @ 2 x 128^3 (each in its own process):
    python tools/gpu_conv_cl_ab.py a.so b.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
from segmamba_amd import lib as L
L.LIB_PATH = sys.argv[1]
from segmamba_amd import ops_raw
hip = L.get_lib()
dev = torch.device("cuda")
B, S = 2, 128
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, S, S, S, 48, device=dev, generator=g).bfloat16()
w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=dev, generator=g)).bfloat16()
img = ops_raw.conv3d_cl_weight_image(hip, w)
out = torch.empty_like(x)
for w8 in (False, True):
    f = lambda: ops_raw.conv3d_k3_fwd_cl(hip, x, img, None, out=out, waves8=w8)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print("    %%s: %%s ms" %% ("8 waves x 2 tiles" if w8 else "4 waves x 4 tiles", " ".join("%%.3f" %% t for t in ts)), flush=True)
''' % ROOT
for so in sys.argv[1:]:
    print("variant", so, flush=True)
    env = dict(os.environ)
    if ":" in so:
        sets, so = so.split(":", 1)
        env.update(kv.split("=", 1) for kv in sets.split(","))
    subprocess.run([sys.executable, "-c", code, os.path.abspath(so)], env=env)
