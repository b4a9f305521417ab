"""segm_skinny_tn against the routes tn_matmul takes without it (segm_wgrad_gemm TN / split-K bmm) at the dt_proj weight-gradient
shapes of the four stages (2 x 4 x 128^3 input): time per call and the HBM rate of the operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw, linear

hip = L.get_lib()
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for K, M, R, ldb in [(524288, 96, 3, 40), (65536, 192, 6, 40), (8192, 384, 12, 48), (1024, 768, 24, 56)]:
    a = torch.randn(K, M, device=dev).bfloat16()
    b = torch.randn(K, ldb, device=dev).bfloat16()[:, :R]
    nbytes = K * (M + R) * 2
    t_sk = timeit(lambda: ops_raw.skinny_tn(hip, a, b))
    linear._SKINNY = False
    t_other = timeit(lambda: linear.tn_matmul(a, b))
    linear._SKINNY = True
    ref = (a.double().t() @ b.double()).float()
    err = float((ops_raw.skinny_tn(hip, a, b) - ref).abs().max() / ref.abs().max())
    print(f"k={K:7d} m={M:4d} n={R:3d}: skinny_tn {t_sk * 1e3:7.1f} us ({nbytes / t_sk / 1e9:6.2f} TB/s)   other route {t_other * 1e3:7.1f} us   rel err {err:.1e}")
