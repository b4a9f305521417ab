"""Time segm_stem_conv_wgrad against ATen's convolution_backward (weight only) at the BASELINE stem shape.
Usage (GPU box): python tools/gpu_stem_wgrad_time.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw

dev = torch.device("cuda")
hip = L.get_lib()
x = torch.rand(2, 4, 128, 128, 128, device=dev).bfloat16()
w = (0.05 * torch.randn(48, 4, 7, 7, 7, device=dev)).bfloat16()
dy = torch.randn(2, 48, 64, 64, 64, device=dev).bfloat16()
x4 = ops_raw.stem_channel_last4(x)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def aten():
    return torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2, 2], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False])[1]


t_hip = timed(lambda: ops_raw.stem_conv_wgrad(hip, x4, dy, 4))
t_aten = timed(aten)
d = (ops_raw.stem_conv_wgrad(hip, x4, dy, 4) - aten().float()).abs().max().item()
print(f"stem wgrad 2x4x128^3 -> 48: library {t_hip:.3f} ms, ATen/MIOpen {t_aten:.3f} ms, max diff {d:.3g}")
