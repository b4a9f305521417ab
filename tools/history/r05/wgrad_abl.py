"""Round 5: where does the weight-gradient row loop spend its time?  Timing ablations of the round-5 kernel (SEGM_WGRAD_ABL, wrong
results by construction), three workgroups per CU (SEGM_WGRAD_OCC=3), items per workgroup; 48 -> 48 @128^3 and 96 -> 96 @64^3."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw

hip = L.get_lib()
ONLY = os.environ.get("WG_ONLY")           # "r5" / "v1": one variant, few launches (counter passes)


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
    for k in ("SEGM_WGRAD_V1", "SEGM_WGRAD_IPW", "SEGM_WGRAD_OCC", "SEGM_WGRAD_ABL"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


variants = [("v1", dict(SEGM_WGRAD_V1=1)), ("r5 ipw1", dict(SEGM_WGRAD_IPW=1)), ("r5 ipw2", dict(SEGM_WGRAD_IPW=2)),
            ("occ3 ipw1", dict(SEGM_WGRAD_OCC=3, SEGM_WGRAD_IPW=1)), ("occ3 ipw2", dict(SEGM_WGRAD_OCC=3, SEGM_WGRAD_IPW=2)),
            ("pipe ipw1", dict(SEGM_WGRAD_OCC=4, SEGM_WGRAD_IPW=1)), ("pipe ipw2", dict(SEGM_WGRAD_OCC=4, SEGM_WGRAD_IPW=2)),
            ("ct ipw1", dict(SEGM_WGRAD_OCC=5, SEGM_WGRAD_IPW=1)), ("ct ipw2", dict(SEGM_WGRAD_OCC=5, SEGM_WGRAD_IPW=2)),
            ("ct ipw4", dict(SEGM_WGRAD_OCC=5, SEGM_WGRAD_IPW=4)),
            ("ct3 ipw1", dict(SEGM_WGRAD_OCC=6, SEGM_WGRAD_IPW=1)), ("ct3 ipw2", dict(SEGM_WGRAD_OCC=6, SEGM_WGRAD_IPW=2))]
if ONLY:
    variants = [v for v in variants if v[0] == ONLY]
for (B, cin, cout, S) in ((2, 48, 48, 128), (2, 96, 48, 128), (2, 96, 96, 64)):
    x = torch.randn(B, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(B, cout, S, S, S, device="cuda").bfloat16()
    flops = 2.0 * B * S ** 3 * cin * cout * 27
    print(f"wgrad B={B} {cin}->{cout} @{S}^3", flush=True)
    for name, env in variants:
        setenv(**env)
        ms = t(lambda: ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.bfloat16), 3 if ONLY else 10)
        print(f"   {name:16s} {ms:.3f} ms ({flops / ms * 1e-9:.0f} TF/s)", flush=True)
setenv()
