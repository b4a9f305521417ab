"""Randomised sweep of the scan / conv1d kernels on the CPU emulation against the oracle (test infrastructure).
    python tools/emu_random_sweep.py [ncases] [seed] [regular]
Draws shapes, layouts, time orders, chunk lengths, dtypes and optional arguments at random; prints the failing
configuration and exits non-zero on the first mismatch."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from tests.emu_util import emu_lib
from segmamba_amd import lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
regular = len(sys.argv) > 3 and sys.argv[3] == "regular"      # bias towards the regular-shape (fast path) kernels
emu = emu_lib()
for case in range(n):
    dim = rng.choice([3, 4, 8, 16, 24, 32, 40, 64, 96, 128])
    dstate = rng.choice([1, 2, 3, 4, 5, 8, 12, 16])
    order = rng.choice([L.TIME_FORWARD, L.TIME_REVERSED, L.TIME_INTERLEAVED])
    ns = rng.choice([1, 2, 3, 4, 8, 16]) if order == L.TIME_INTERLEAVED else 1
    base = rng.choice([8, 16, 24, 40, 48, 64, 96, 100, 128, 160])
    seqlen = max(ns, (base // ns) * ns)
    chunk = rng.choice([16, 32, 48, 64])
    channel_last = rng.random() < 0.6
    groups = rng.choice([1, 1, 1, 2]) if dim % 2 == 0 else 1
    dtype = rng.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
    has_z, has_D, has_bias = rng.random() < 0.8, rng.random() < 0.8, rng.random() < 0.8
    softplus = rng.random() < 0.8
    batch = rng.choice([1, 2])
    if regular:
        dstate = 16
        dim = rng.choice([16, 32, 48, 64, 96, 128])
        chunk = rng.choice([16, 32, 64])
        ns = rng.choice([8, 16]) if order == L.TIME_INTERLEAVED else 1
        if order == L.TIME_INTERLEAVED:
            chunk = max(chunk, ns)
        seqlen = chunk * rng.choice([1, 2, 4, 6])
        groups = 1
    cfg = dict(batch=batch, dim=dim, dstate=dstate, seqlen=seqlen, order=order, ns=ns, chunk=chunk, channel_last=channel_last,
               groups=groups, dtype=str(dtype), has_z=has_z, has_D=has_D, has_bias=has_bias, softplus=softplus)
    try:
        c = H.scan_case(batch, dim, dstate, seqlen, groups=groups, dtype=dtype, has_z=has_z, has_D=has_D, has_bias=has_bias,
                        seed=case + 1000)
        ref = H.scan_oracle(c, order, ns, softplus=softplus)
        res = H.run_scan(emu, c, "cpu", channel_last, order, ns, chunk=chunk, softplus=softplus)
        H.check_scan(res, ref, dtype, f"case {case}")
    except Exception as e:                                   # noqa: BLE001 - report the configuration, then fail
        print("FAILED", cfg, "\n", repr(e)[:600])
        sys.exit(1)
    print(f"ok {case:3d} {cfg}", flush=True)
print("all", n, "cases agree with the oracle")
