"""segm_linear_rows with and without `accumulate` at the Mamba backward's shapes (dconv += dx_dbl W_x: rows x 40 -> 96 / 192)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu
hip = L.get_lib()
dev = torch.device("cuda")
for rows, K, N in ((524288, 40, 96), (65536, 40, 192), (524288, 96, 192), (524288, 96, 48)):
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
