"""GPU timing of segm_conv3d_k3_wgrad against MIOpen's weight gradient on the SegMamba stem / decoder shapes."""
import os, sys
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw

hip = L.get_lib()
dev = "cuda"


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for (B, cin, cout, S) in ((2, 48, 48, 128), (2, 96, 96, 64), (2, 48, 48, 64)):
    x = torch.randn(B, cin, S, S, S, device=dev).bfloat16()
    dy = torch.randn(B, cout, S, S, S, device=dev).bfloat16()
    w = torch.empty(cout, cin, 3, 3, 3, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.bfloat16))
    flops = 2.0 * B * S ** 3 * cin * cout * 27
    ms_ref = t(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1,
                                                           [False, True, False])[1], 2)
    ref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w.float(), None, [1] * 3, [1] * 3, [1] * 3, False,
                                              [0] * 3, 1, [False, True, False])[1]
    got = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"wgrad B={B} {cin}->{cout} @{S}^3: mfma {ms:.3f} ms ({flops / ms * 1e-9:.1f} TF/s, "
          f"{(x.numel() + dy.numel()) * 2 / ms * 1e-6:.0f} GB/s), MIOpen {ms_ref:.3f} ms, rel err {err:.2e}", flush=True)

for (B, cout, S) in ((2, 48, 128), (2, 48, 64), (2, 96, 64), (2, 48, 32)):
    x = torch.randn(B, 48, S, S, S, device=dev).bfloat16()
    w = (0.05 * torch.randn(cout, 48, 3, 3, 3, device=dev)).bfloat16()
    wp = ops_raw.pack_conv3d_weight(w)
    ms = t(lambda: ops_raw.conv3d_k3_fwd(hip, x, wp))
    ms_ref = t(lambda: torch.nn.functional.conv3d(x, w, None, 1, 1), 3)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    err = ((ops_raw.conv3d_k3_fwd(hip, x, wp).float() - ref).abs().max() / ref.abs().max()).item()
    flops = 2.0 * B * S ** 3 * 48 * cout * 27
    print(f"fwd B={B} 48->{cout} @{S}^3: hip {ms:.3f} ms ({flops / ms * 1e-9:.1f} TF/s), MIOpen {ms_ref:.3f} ms, rel err {err:.2e}",
          flush=True)
