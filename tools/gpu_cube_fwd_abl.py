"""timing of segm_conv3d_k3_cube_fwd build variants per layer and plan: python tools/gpu_cube_fwd_abl.py lib1.so lib2.so ..."""
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
CASES = [(768, 384, 16, 3, 4), (384, 384, 16, 3, 4), (384, 384, 16, 2, 2), (192, 192, 16, 2, 3), (384, 192, 32, 3, 1), (192, 192, 32, 3, 1), (192, 192, 32, 2, 1), (768, 768, 8, 2, 8), (768, 768, 8, 3, 12)]
for path in sys.argv[1:]:
    lib = L.SegmLib(os.path.abspath(path))
    out = "%-28s" % os.path.basename(path)
    for cin, cout, S, nt, sp in CASES:
        x = torch.randn(2, cin, S, S, S, device="cuda").bfloat16()
        w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device="cuda")).bfloat16()
        img = ops_raw.conv3d_cube_weight_image(lib, w)
        ms = t(lambda: ops_raw.conv3d_k3_cube_fwd(lib, x, img, cout, nt=nt, splits=sp))
        out += " %d>%d@%d nt%d s%d %.3f |" % (cin, cout, S, nt, sp, ms)
    print(out, flush=True)
