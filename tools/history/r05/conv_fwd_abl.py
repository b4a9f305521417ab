"""Round 5: where does the chained 3x3x3 forward kernel (64-wide, 48 -> 48) spend its row step?  Timing ablations (SEGM_CONV_FWD_ABL:
wrong results by construction), 48 -> 48 @128^3 x 2 and 96 -> 96 @64^3 on the 64-wide kernel."""
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


names = {0: "shipped", 1: "no row fetch", 2: "no MFMA", 3: "no fragment reads", 4: "no park", 5: "no barrier", 6: "no hand-off tiles", 7: "no output store"}
for (B, c, S) in ((2, 48, 128), (2, 48, 64)):
    x = torch.randn(B, c, S, S, S, device="cuda").bfloat16()
    w = (0.05 * torch.randn(c, c, 3, 3, 3, device="cuda")).bfloat16()
    wp = ops_raw.pack_conv3d_weight(w)
    fl = 2.0 * B * S ** 3 * c * c * 27
    print(f"conv fwd {c}->{c} @{S}^3 B={B} (64-wide chained kernel)", flush=True)
    for rep in range(2):
        for abl in range(8):
            os.environ["SEGM_CONV_FWD_ABL"] = str(abl)
            ms = t(lambda: ops_raw.conv3d_k3_fwd(hip, x, wp, None, chain=True, pitch48=True))
            print(f"   {names[abl]:20s} {ms:.4f} ms ({fl / ms * 1e-9:.0f} TF/s)", flush=True)
os.environ["SEGM_CONV_FWD_ABL"] = "0"
