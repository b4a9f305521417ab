"""Round 5: segm_conv3d_k3_wgrad, the round-5 row loop against the round-1..4 kernel (SEGM_WGRAD_V1=1) and items-per-workgroup
settings (SEGM_WGRAD_IPW), on the shapes of the training step.  Both environment switches are read per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw

hip = L.get_lib()


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


def setenv(**kw):
    for k in ("SEGM_WGRAD_V1", "SEGM_WGRAD_IPW", "SEGM_WGRAD_DIRECT"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


for (B, cin, cout, S) in ((2, 48, 48, 128), (2, 96, 48, 128), (2, 96, 96, 64), (2, 192, 96, 64), (2, 192, 192, 32), (2, 384, 384, 16)):
    x = torch.randn(B, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(B, cout, S, S, S, device="cuda").bfloat16()
    flops = 2.0 * B * S ** 3 * cin * cout * 27
    setenv(SEGM_WGRAD_V1=1)
    ref = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
    line = f"wgrad B={B} {cin}->{cout} @{S}^3:"
    for name, env in (("v1", dict(SEGM_WGRAD_V1=1)), ("r5", {}), ("r5 ipw1", dict(SEGM_WGRAD_IPW=1)), ("r5 ipw2", dict(SEGM_WGRAD_IPW=2)),
                      ("r5 ipw4", dict(SEGM_WGRAD_IPW=4))):
        setenv(**env)
        got = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        ms = t(lambda: ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.bfloat16))
        line += f"  {name} {ms:.3f} ms ({flops / ms * 1e-9:.0f} TF/s, vs v1 {err:.1e})"
    print(line, flush=True)
setenv()
