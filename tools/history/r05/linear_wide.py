"""Round 5: segm_linear_rows with 16-byte stores (tile pairs, SEGM_LINEAR_WIDE=1, the default) against the 8-byte form, at the Mamba
block's projection shapes of stages 0 / 1 (rows = B L).  One process per setting (the switch is read once)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu
hip = L.get_lib()
dev = torch.device("cuda")
print("SEGM_LINEAR_WIDE =", os.environ.get("SEGM_LINEAR_WIDE", "1"))
for rows, K, N in ((524288, 48, 192), (524288, 96, 48), (524288, 96, 40), (524288, 40, 96), (524288, 8, 96), (524288, 192, 48), (524288, 48, 96),
                   (65536, 96, 384), (65536, 192, 96), (65536, 384 // 2, 96)):
    x = torch.randn(rows, K, device=dev).bfloat16()
    w = (0.1 * torch.randn(N, K, device=dev)).bfloat16()
    y = torch.randn(rows, N, device=dev).bfloat16()
    t0 = time_gpu(lambda: ops_raw.linear_rows(hip, x, w, out=y), 20)
    t1 = time_gpu(lambda: ops_raw.linear_rows(hip, x, w, out=y, accumulate=True), 20)
    y0 = torch.zeros_like(y); ops_raw.linear_rows(hip, x, w, out=y0, accumulate=True)
    ref = x.float() @ w.float().t()
    err = float((y0.float() - ref).abs().max() / ref.abs().max())
    gb = (rows * K + rows * N) * 2 / 1e9
    print(f"rows {rows} K {K} N {N}: plain {t0 * 1e3:6.1f} us ({gb / t0 * 1e3:.0f} GB/s)  accumulate {t1 * 1e3:6.1f} us ({(gb + rows * N * 2 / 1e9) / t1 * 1e3:.0f} GB/s)  rel err {err:.1e}", flush=True)
