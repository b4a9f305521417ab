"""A/B timing of library variants on the selective-scan roofline shape (stage 0: B=2, D=96, N=16, L=64^3, channel-last):
    python tools/gpu_scan_ab.py build/variants/a.so build/variants/b.so ...
Each variant runs in its own process; prints forward / backward ms (HIP events) and, with --check, the max error of the
forward output and three gradients against the first variant on the same inputs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
from segmamba_amd import lib as L
L.LIB_PATH = sys.argv[1]
from bench import scan_roofline
import bench
from segmamba_amd import ops_raw
hip = L.get_lib()
for dt in (torch.bfloat16, torch.float32):
    r = scan_roofline(dt, torch.device("cuda"))
    t3 = r.get("three_directions_per_launch") or {}
    print("    %%-8s chunk %%d  fwd %%.4f ms (frac %%.3f)  bwd %%.4f ms   three directions per launch: fwd %%s ms  bwd %%s ms" %% (
        str(dt).split(".")[1], r["shape"]["chunk"], r["ms"], r["frac"], r["backward"]["ms"], t3.get("fwd_ms"), t3.get("bwd_ms")), flush=True)
# fingerprint of the results (same seeded inputs in every variant)
g = torch.Generator(device="cuda").manual_seed(1)
B, D, N, Lq = 2, 96, 16, 8192
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
delta = 0.5 * torch.rand(B, Lq, D, device="cuda", generator=g)
A = -0.5 * torch.rand(D, N, device="cuda", generator=g); Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
Dv = torch.randn(D, device="cuda", generator=g); db = 0.5 * torch.rand(D, device="cuda", generator=g)
f = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True)
b = ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True, chunk=f["chunk"])
torch.save({"out_z": f["out_z"].cpu(), "du": b["du"].cpu(), "dB": b["dB"].cpu(), "dA": b["dA"].cpu()}, sys.argv[2])
''' % ROOT
outs = []
for i, so in enumerate(a for a in sys.argv[1:] if not a.startswith("--")):
    print("variant", so, flush=True)                      # "NAME=VALUE,NAME=VALUE:path.so" runs the library with that environment
    env = dict(os.environ)
    if ":" in so:
        sets, so = so.split(":", 1)
        env.update(kv.split("=", 1) for kv in sets.split(","))
    out = f"/tmp/scan_ab_{i}.pt"
    subprocess.run([sys.executable, "-c", code, os.path.abspath(so), out], env=env)
    outs.append(out)
if len(outs) > 1:
    import torch
    ref = torch.load(outs[0])
    for o in outs[1:]:
        if not os.path.exists(o):
            continue
        t = torch.load(o)
        print("vs first:", {k: "%.2e" % float((t[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref})
