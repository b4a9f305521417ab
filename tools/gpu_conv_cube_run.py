"""a few launches of the cube kernels at fixed shapes (for rocprofv3 --pmc passes: tools/gpu_conv_cube_pmc.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
for cin, cout, S in ((768, 384, 16), (384, 192, 32), (768, 768, 8)):
    x = torch.randn(2, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(2, cout, S, S, S, device="cuda").bfloat16()
    w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device="cuda")).bfloat16()
    img = ops_raw.conv3d_cube_weight_image(hip, w)
    for _ in range(4):
        ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout)
        ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, torch.float32)
torch.cuda.synchronize()
