"""segm_conv3d_k3_cube_wgrad on the wide layers of the 8^3 / 16^3 / 32^3 levels: parity against fp32 ATen and time next to the product's
present weight-gradient route.      python tools/gpu_conv_cube_wgrad_time.py [out.txt]"""
import os, sys
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import conv3d as C, lib as L, ops_raw

hip = L.get_lib()
dev = "cuda"
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


LAYERS = [(192, 192, 16), (192, 384, 16), (384, 384, 16), (768, 384, 16), (384, 384, 8), (384, 768, 8), (768, 768, 8),
          (96, 192, 32), (192, 192, 32), (384, 192, 32)]
B = 2
os.environ["SEGM_CONV_CUBE_WGRAD"] = "0"
for cin, cout, S in LAYERS:
    x = torch.randn(B, cin, S, S, S, device=dev).bfloat16()
    dy = torch.randn(B, cout, S, S, S, device=dev).bfloat16()
    w = torch.zeros(cout, cin, 3, 3, 3, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * B * S ** 3 * cin * cout * 27
    ref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w.float(), None, [1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1, [False, True, False])[1]
    C._CUBE_WGRAD = False
    ms0 = t(lambda: C._wgrad(x, dy, w, 1, torch.float32))
    dw = ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, torch.float32)
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    ms = t(lambda: ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, torch.float32))
    say("%3d -> %3d @%2d^3: present route %.3f ms (%4.0f TF/s) | cube wgrad %.3f ms (%4.0f TF/s, err %.1e) = %.2fx" % (
        cin, cout, S, ms0, fl / ms0 * 1e-9, ms, fl / ms * 1e-9, err, ms0 / ms))
    del x, dy, w, ref, dw
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
