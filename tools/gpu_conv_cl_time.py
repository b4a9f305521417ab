"""The channel-last 3x3x3 forward prototype (csrc/conv3d_cl.hip) against the shipped NCDHW kernel and fp32 ATen:
    python tools/gpu_conv_cl_time.py [out.log]
48 -> 48 at 2 x 128^3 (the benchmarked layer shape), 2 x 64^3, 2 x 32^3; both wave arrangements; accumulate (the second block
of a 96 -> 48 layer).  Error = max |y - fp32 conv of the same bf16 operands| / max |ref| (bound of the at-size test: 1e-2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmamba_amd import lib as L, ops_raw  # noqa: E402

hip = L.get_lib()
dev = torch.device("cuda")
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


def time_ms(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


say("device:", torch.cuda.get_device_name(0))
for (B, S) in ((2, 128), (2, 64), (2, 32)):
    g = torch.Generator(device=dev).manual_seed(S)
    x = torch.randn(B, 48, S, S, S, device=dev, generator=g).bfloat16()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=dev, generator=g)).bfloat16()
    flop = 2.0 * B * S ** 3 * 27 * 48 * 48
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    scale = float(ref.abs().max())
    # row pitch padded by PADX voxels: a dense 128-voxel row is 12 KB = 3 x 4096 bytes and the 16 rows a fragment load touches would
    # all sit on one memory channel (the same aliasing the NCDHW volumes avoid with a padded channel stride)
    PADX = int(os.environ.get("SEGM_CL_PADX", "0"))
    xcl = torch.zeros(B, S, S, S + PADX, 48, dtype=x.dtype, device=dev)[:, :, :, :S]
    xcl.copy_(x.permute(0, 2, 3, 4, 1))
    img = ops_raw.conv3d_cl_weight_image(hip, w)
    # shipped NCDHW kernel (padded channel stride as the step uses it at 128^3)
    xp = ops_raw.volume_empty(B, 48, (S, S, S), x.dtype, dev)
    xp.copy_(x)
    wp = ops_raw.pack_conv3d_weight(w)
    kw = dict(chain=True, pitch48=True) if S >= 64 else dict(chain32=True)
    y0 = ops_raw.conv3d_k3_fwd(hip, xp, wp, None, **kw)
    e0 = float((y0.float() - ref).abs().max()) / scale
    t0 = time_ms(lambda: ops_raw.conv3d_k3_fwd(hip, xp, wp, None, **kw))
    say("48 -> 48 @ %d x %d^3   NCDHW shipped (%s): %.3f ms = %.0f TF/s   err %.2e" % (B, S, "chain64" if S >= 64 else "chain32", t0, flop / t0 * 1e-9, e0))
    for w8 in (False, True):
        y = ops_raw.conv3d_k3_fwd_cl(hip, xcl, img, None, waves8=w8)
        e = float((y.permute(0, 4, 1, 2, 3).float() - ref).abs().max()) / scale
        out = torch.zeros(B, S, S, S + PADX, 48, dtype=x.dtype, device=dev)[:, :, :, :S]
        t = time_ms(lambda: ops_raw.conv3d_k3_fwd_cl(hip, xcl, img, None, out=out, waves8=w8))
        ta = time_ms(lambda: ops_raw.conv3d_k3_fwd_cl(hip, xcl, img, None, out=out, accumulate=True, waves8=w8))
        say("                       channel-last %s: %.3f ms = %.0f TF/s (%.1f %% of 2.5 PF)   err %.2e   accumulate %.3f ms" % (
            "8 waves x 2 tiles" if w8 else "4 waves x 4 tiles", t, flop / t * 1e-9, flop / t * 1e-9 / 25.0, e, ta))
    del ref
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
