"""Where a wave of the channel-last 3x3x3 kernel spends its cycles (build with -DSEGM_CL_TIMELINE):
    SEGM_LIB_OUT=build/variants/libsegm_cl_tl.so SEGM_EXTRA_HIPCC_FLAGS=-DSEGM_CL_TIMELINE python -m segmamba_amd.build
    python tools/gpu_conv_cl_timeline.py build/variants/libsegm_cl_tl.so [out.txt]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmamba_amd import lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
from segmamba_amd import ops_raw  # noqa: E402

hip = L.get_lib()
dev = torch.device("cuda")
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


B, S = 2, 128
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, S, S, S, 48, device=dev, generator=g).bfloat16()
w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=dev, generator=g)).bfloat16()
img = ops_raw.conv3d_cl_weight_image(hip, w)
out = torch.empty_like(x)
dbg = hip.dll.segm_debug_set_cl_timeline
dbg.argtypes = [ctypes.c_void_p]
dbg.restype = ctypes.c_int
for w8 in (True, False):
    waves = 8 if w8 else 4
    for _ in range(3):
        ops_raw.conv3d_k3_fwd_cl(hip, x, img, None, out=out, waves8=w8)
    buf = torch.zeros(256 * waves * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert dbg(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops_raw.conv3d_k3_fwd_cl(hip, x, img, None, out=out, waves8=w8)
    e1.record()
    torch.cuda.synchronize()
    assert dbg(0) == 0
    r = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    r = r[r[:, 0] > 0]
    tot, k, e, n, p = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
    hw = r[:, 7].astype(np.int64)
    simd = (hw >> 4) & 3
    say("%d waves x %d tiles: %.3f ms; %d waves reported; per wave: lifetime %.0f cycles (min %.0f max %.0f), prologue (weights -> LDS) %.0f" % (
        waves, 2 if w8 else 4, e0.elapsed_time(e1), len(r), tot.mean(), tot.min(), tot.max(), p.mean()))
    mf = 126 * (2 if w8 else 4)
    say("   groups per wave %.1f; k loop %.0f cycles per group (%d MFMAs = %d cycles of matrix pipe; min %.0f max %.0f over waves), epilogue %.0f (min %.0f max %.0f); "
        "k loops %.1f %% + epilogues %.1f %% + rest %.1f %% of the lifetime" % (
            n.mean(), (k / n).mean(), mf, mf * 16, (k / n).min(), (k / n).max(), (e / n).mean(), (e / n).min(), (e / n).max(),
            100 * k.sum() / tot.sum(), 100 * e.sum() / tot.sum(), 100 * (tot - k - e).sum() / tot.sum()))
    # the two (or one) waves of a SIMD: order by lifetime within (block, simd)
    span = r[:, 6].max() - r[:, 5].min()
    say("   launch span %.0f cycles (first entry -> last exit, s_memtime); wave end times (cycles after first entry): p10 %.0f p50 %.0f p90 %.0f max %.0f" % (
        span, *np.percentile(r[:, 6] - r[:, 5].min(), [10, 50, 90, 100])))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
