"""Round 5: segm_pointwise_cf with 16-byte stores (voxel blocks dealt in pairs, SEGM_POINTWISE_WIDE=1, the default) against the 8-byte
form at the stem's 1x1x1 shapes, on padded volumes as in the step.  One process per setting (the switch is read once)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu
hip = L.get_lib()
print("SEGM_POINTWISE_WIDE =", os.environ.get("SEGM_POINTWISE_WIDE", "1"))
for B, Cin, Cout, S in [(2, 48, 48, 128 ** 3), (2, 4, 48, 128 ** 3), (2, 48, 4, 128 ** 3), (2, 48, 96, 64 ** 3), (2, 96, 48, 64 ** 3), (2, 96, 96, 64 ** 3), (2, 96, 96, 32 ** 3)]:
    x = ops_raw.volume_empty(B, Cin, (S,), torch.bfloat16, "cuda"); x.copy_(torch.randn(B, Cin, S, device="cuda"))
    w = (0.1 * torch.randn(Cout, Cin, device="cuda")).bfloat16()
    b = torch.randn(Cout, device="cuda")
    y = ops_raw.pointwise_cf(hip, x, w, b)
    t0 = time_gpu(lambda: ops_raw.pointwise_cf(hip, x, w, b, out=y), 20)
    t1 = time_gpu(lambda: ops_raw.pointwise_cf(hip, x, w, None, out=y, accumulate=True), 20)
    y0 = ops_raw.pointwise_cf(hip, x, w, b)
    ref = torch.einsum("oc,bcs->bos", w.float(), x[:, :, :65536].float()) + b.view(1, -1, 1)
    err = float((y0[:, :, :65536].float() - ref).abs().max() / ref.abs().max())
    gb = B * S * (Cin + Cout) * 2 / 1e9
    print(f"pointwise {Cin}->{Cout} S={S}: plain {t0 * 1e3:6.1f} us ({gb / t0 * 1e3:.0f} GB/s)  accumulate {t1 * 1e3:6.1f} us ({(gb + B * S * Cout * 2 / 1e9) / t1 * 1e3:.0f} GB/s)  rel err {err:.1e}", flush=True)
