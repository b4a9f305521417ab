"""Does the power-of-two channel stride of NCDHW volumes (128^3 x 2 B = 4 MiB) cost the conv kernels bandwidth (every channel
of a row in the same L2 set / memory channel)?  Times segm_conv3d_k3_fwd (chained, pitch 48) on the same data with the channel
stride padded by `pad` elements, for input and output separately.  Usage (GPU box): python tools/gpu_conv_stride_pad.py"""
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


def padded(B, C, S, pad):
    buf = torch.zeros(B, C, S ** 3 + pad, device=dev, dtype=torch.bfloat16)
    return buf[:, :, :S ** 3].view(B, C, S, S, S)


for (B, C, S) in [(2, 48, 128), (2, 48, 64)]:
    x0 = torch.randn(B, C, S, S, S, device=dev).bfloat16()
    w = (0.05 * torch.randn(C, C, 3, 3, 3, device=dev)).bfloat16()
    wp = ops_raw.pack_conv3d_weight(w)
    ref = ops_raw.conv3d_k3_fwd(hip, x0, wp, None, chain=True, pitch48=True)
    for padx, pady in [(0, 0), (64, 0), (0, 64), (64, 64), (192, 192), (1088, 1088), (8256, 8256)]:
        x = padded(B, C, S, padx)
        x.copy_(x0)
        y = padded(B, C, S, pady)
        t = timeit(lambda: ops_raw.conv3d_k3_fwd(hip, x, wp, None, out=y, chain=True, pitch48=True))
        d = (y.float() - ref.float()).abs().max().item()
        fl = 2.0 * B * S ** 3 * C * C * 27
        print(f"conv {C}->{C} @{S}^3 B={B}: channel stride pad x {padx:5d} y {pady:5d}: {t:.3f} ms ({fl / t / 1e9:.0f} TF/s) maxdiff {d:.2g}", flush=True)

# the weight-gradient kernel and the 1x1x1 kernel on the same padded volumes
B, C, S = 2, 48, 128
x0 = torch.randn(B, C, S, S, S, device=dev).bfloat16()
g0 = torch.randn(B, C, S, S, S, device=dev).bfloat16()
w2 = (0.05 * torch.randn(C, C, device=dev)).bfloat16()
for pad in (0, 192):
    x, g = padded(B, C, S, pad), padded(B, C, S, pad)
    x.copy_(x0)
    g.copy_(g0)
    t = timeit(lambda: ops_raw.conv3d_k3_wgrad(hip, x, g, torch.float32))
    print(f"wgrad 48x48 @128^3 B=2: channel stride pad {pad}: {t:.3f} ms", flush=True)
    y = padded(B, C, S, pad)
    t = timeit(lambda: ops_raw.pointwise_cf(hip, x.view(B, C, S ** 3), w2, None, out=y.view(B, C, S ** 3)))
    print(f"pointwise 48->48 @128^3 B=2: channel stride pad {pad}: {t:.3f} ms", flush=True)
