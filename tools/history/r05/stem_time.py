"""Round 5: the thin-input convolution kernels (7^3 stride-2 stem, 3^3 first layer) before / after the two-deep load pipeline and the
paired 16-byte stores (SEGM_STEM_WIDE per launch; the pipeline is not switchable - compare with profiles/r05_step_kernels_final.txt:
0.392 / 0.173 ms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu
hip = L.get_lib()
x = torch.rand(2, 4, 128, 128, 128, device="cuda").bfloat16()
x4 = ops_raw.stem_channel_last4(x)                        # (the transposing copy of the input is made once per step and shared)
for k, name in ((3, "3^3 s1 4->48 @128^3"), (7, "7^3 s2 4->48 @128^3 -> 64^3")):
    w = (0.05 * torch.randn(48, 4, k, k, k, device="cuda")).bfloat16()
    b = torch.randn(48, device="cuda")
    ref = torch.nn.functional.conv3d(x[:, :, :16].float(), w.float(), b, stride=1 if k == 3 else 2, padding=k // 2)
    for wide in ("0", "1"):
        os.environ["SEGM_STEM_WIDE"] = wide
        y = ops_raw.stem_conv_fwd(hip, x, w, b, x4=x4)
        nz = ref.shape[2] - 2
        err = float((y[:, :, :nz].float() - ref[:, :, :nz]).abs().max() / ref.abs().max())
        t = time_gpu(lambda: ops_raw.stem_conv_fwd(hip, x, w, b, x4=x4), 20)
        print(f"{name}: SEGM_STEM_WIDE={wide} {t * 1e3:6.1f} us  ({(x.numel() + y.numel()) * 2 / t / 1e9:.2f} TB/s of input + output)  rel err {err:.1e}", flush=True)
