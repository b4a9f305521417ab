"""segm_linear_rows against torch's GEMM on the projection shapes of stages 2 / 3 (a few thousand rows, K up to 768)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
dev = "cuda"


def t_us(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, K, N, what in [(8192, 192, 768, "in_proj s2"), (8192, 384, 192, "out_proj s2"), (8192, 384, 48, "x_proj s2"), (8192, 16, 384, "dt_proj s2"),
                         (8192, 768, 192, "in_proj dgrad s2"), (1024, 384, 1536, "in_proj s3"), (1024, 768, 384, "out_proj s3"), (1024, 768, 56, "x_proj s3"),
                         (1024, 24, 768, "dt_proj s3"), (65536, 192, 96, "out_proj s1 (stationary-W kernel)")]:
    x = torch.randn(rows, K, device=dev).bfloat16()
    w = (0.1 * torch.randn(N, K, device=dev)).bfloat16()
    y = ops_raw.linear_rows(hip, x, w)
    ref = x.float() @ w.float().t()
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print("%-34s rows %6d K %4d N %4d: library %6.1f us   torch mm %6.1f us   err %.1e" % (
        what, rows, K, N, t_us(lambda: ops_raw.linear_rows(hip, x, w)), t_us(lambda: torch.mm(x, w.t())), err), flush=True)
