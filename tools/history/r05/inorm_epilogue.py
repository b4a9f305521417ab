"""Round 5: what do InstanceNorm statistics cost in the convolution epilogue?  MEASURED, not estimated: the 64-wide
chained 3x3x3 kernel with and without the per-channel sum / sum-of-squares of what its storing K part writes
(SEGM_CONV_STATS_PROBE=1: partials to a scratch buffer), next to the statistics pass they would replace (segm_instnorm_fwd's first
launch, timed through ops_raw.instnorm_fwd minus its apply pass is not separable here - the whole forward is printed and the
statistics kernel's share comes from the step profile: 52 launches, 1.08 ms per step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw

hip = L.get_lib()


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for (B, cin, cout, S) in ((2, 48, 48, 128), (2, 96, 48, 128), (2, 96, 96, 64)):
    x = torch.randn(B, cin, S, S, S, device="cuda").bfloat16()
    w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device="cuda")).bfloat16()
    wps = [ops_raw.pack_conv3d_weight(w[:, i:i + 48]) for i in range(0, cin, 48)]

    def run():
        out = None
        for i, wp in enumerate(wps):
            out = ops_raw.conv3d_k3_fwd(hip, x[:, 48 * i:48 * i + 48], wp, None, out=out, accumulate=i > 0, chain=True, pitch48=True)
        return out
    res = {}
    for rep in range(2):
        for probe in (0, 1):
            os.environ["SEGM_CONV_STATS_PROBE"] = str(probe)
            res.setdefault(probe, []).append(t(run))
    os.environ["SEGM_CONV_STATS_PROBE"] = "0"
    y = run()
    ms_in = t(lambda: ops_raw.instnorm_fwd(hip, y, None, "leaky_relu", 0.01, 1e-5))
    print(f"conv {cin}->{cout} @{S}^3: plain {min(res[0]):.4f} ms  with statistics epilogue {min(res[1]):.4f} ms  "
          f"(delta {min(res[1]) - min(res[0]):+.4f} ms; runs {['%.4f' % v for v in res[0]]} / {['%.4f' % v for v in res[1]]})  |  "
          f"instnorm_fwd (statistics + apply) on its output {ms_in:.4f} ms", flush=True)
