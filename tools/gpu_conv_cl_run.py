"""Runs the channel-last 3x3x3 forward kernel a few times at 48 -> 48 @ 2 x 128^3 (for rocprofv3 counter passes).
    python tools/gpu_conv_cl_run.py [w8|w4|both]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmamba_amd import lib as L, ops_raw  # noqa: E402

hip = L.get_lib()
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
B, S = 2, 128
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, S, S, S, 48, device=dev, generator=g).bfloat16()
w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=dev, generator=g)).bfloat16()
img = ops_raw.conv3d_cl_weight_image(hip, w)
out = torch.empty_like(x)
for w8 in ((True,) if which == "w8" else (False,) if which == "w4" else (False, True)):
    for _ in range(4):
        ops_raw.conv3d_k3_fwd_cl(hip, x, img, None, out=out, waves8=w8)
torch.cuda.synchronize()
