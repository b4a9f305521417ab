"""The network's first 3x3x3 layer (4 -> 48 at 2 x 128^3): thin-input kernels against the 48-channel kernels (44 zero channels).
Usage (GPU box): python tools/gpu_thin_conv_time.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw

dev = "cuda:0"
hip = L.get_lib()


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.rand(2, 4, 128, 128, 128, device=dev).bfloat16()
w = (0.1 * torch.randn(48, 4, 3, 3, 3, device=dev)).bfloat16()
dy = torch.randn(2, 48, 128, 128, 128, device=dev).bfloat16()
x4 = ops_raw.stem_channel_last4(x)
wp = ops_raw.pack_conv3d_weight(w)
print(f"forward  4->48 @128^3 x2: thin {timeit(lambda: ops_raw.stem_conv_fwd(hip, x, w, None, x4=x4)):.3f} ms (+ channel-last copy "
      f"{timeit(lambda: ops_raw.stem_channel_last4(x)):.3f} ms), 48-channel kernel {timeit(lambda: ops_raw.conv3d_k3_fwd(hip, x, wp, None, chain=True, pitch48=True)):.3f} ms")
print(f"wgrad    4->48 @128^3 x2: thin {timeit(lambda: ops_raw.stem_conv_wgrad(hip, x4, dy, 4, 3)):.3f} ms, 48-channel kernel "
      f"{timeit(lambda: ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)):.3f} ms")
