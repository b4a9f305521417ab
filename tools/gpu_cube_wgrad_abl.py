"""timing of segm_conv3d_k3_cube_wgrad variants (ablation builds): python tools/gpu_cube_wgrad_abl.py lib1.so lib2.so ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for path in sys.argv[1:]:
    lib = L.SegmLib(os.path.abspath(path))
    out = os.path.basename(path) + ":"
    for cin, cout, S in ((384, 384, 16), (768, 768, 8), (384, 192, 32)):
        x = torch.randn(2, cin, S, S, S, device="cuda").bfloat16(); dy = torch.randn(2, cout, S, S, S, device="cuda").bfloat16()
        out += "  %d->%d@%d %.3f ms" % (cin, cout, S, t(lambda: ops_raw.conv3d_k3_cube_wgrad(lib, x, dy, torch.float32)))
    print(out, flush=True)
