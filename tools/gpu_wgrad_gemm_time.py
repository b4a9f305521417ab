"""segm_wgrad_gemm against the slab-batched BLAS route (linear.tn_matmul / nt_matmul_rows with SEGM_WGRAD_GEMM_HIP=0) on the
weight-gradient shapes of one SegMamba training step (2 x 4 x 128^3).  Usage (GPU box): python tools/gpu_wgrad_gemm_time.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw, linear as LN

dev = "cuda:0"
hip = L.get_lib()


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


LN._WG_TN = True                     # time the opt-in TN kernel too


def blas(fn, *args):
    LN._WG_HIP = False
    try:
        return fn(*args)
    finally:
        LN._WG_HIP = True


print("TN: a (K, M)^T b (K, N)")
for K, M, N, lda, ldb, what in [(524288, 192, 48, 192, 48, "in_proj s0"), (524288, 48, 96, 48, 96, "out_proj s0"),
                                (524288, 35, 96, 35, 96, "x_proj s0"), (524288, 36, 96, 40, 96, "x_proj s0 (40-col)"), (524288, 96, 3, 96, 35, "dt_proj s0"),
                                (65536, 384, 96, 384, 96, "in_proj s1"), (65536, 96, 192, 96, 192, "out_proj s1"),
                                (65536, 38, 192, 38, 192, "x_proj s1"), (65536, 40, 192, 40, 192, "x_proj s1 (40-col)"), (65536, 192, 6, 192, 38, "dt_proj s1"),
                                (8192, 768, 192, 768, 192, "in_proj s2"), (8192, 192, 384, 192, 384, "out_proj s2"), (8192, 44, 384, 48, 384, "x_proj s2"),
                                (1024, 1536, 384, 1536, 384, "in_proj s3"), (1024, 384, 768, 384, 768, "out_proj s3")]:
    a = torch.randn(K, lda, device=dev).bfloat16()[:, :M]
    b = torch.randn(K, ldb, device=dev).bfloat16()[:, :N]
    t1 = timeit(lambda: LN.tn_matmul(a, b))
    t0 = timeit(lambda: blas(LN.tn_matmul, a, b))
    d = (LN.tn_matmul(a, b) - blas(LN.tn_matmul, a, b)).abs().max().item()
    gb = K * (M + N) * 2 / 1e9
    print(f"  {what:12s} K={K} {M}x{N}: library {t1 * 1e3:7.1f} us ({gb / t1:5.2f} TB/s)   BLAS slabs {t0 * 1e3:7.1f} us ({gb / t0:5.2f} TB/s)   maxdiff {d:.3g}", flush=True)
print("NT: sum_b a[b] (M, K) b[b] (N, K)^T")
for Bn, M, N, K, what in [(2, 48, 48, 128 ** 3, "conv3 48->48 @128^3"), (2, 48, 4, 128 ** 3, "conv3 4->48 @128^3"), (2, 4, 48, 128 ** 3, "out 48->4 @128^3"),
                          (2, 96, 96, 64 ** 3, "96x96 @64^3"), (2, 96, 48, 64 ** 3, "96x48 @64^3")]:
    a = torch.randn(Bn, M, K, device=dev).bfloat16()
    b = torch.randn(Bn, N, K, device=dev).bfloat16()
    t1 = timeit(lambda: LN.nt_matmul_rows(a, b))
    t0 = timeit(lambda: blas(LN.nt_matmul_rows, a, b))
    d = (LN.nt_matmul_rows(a, b) - blas(LN.nt_matmul_rows, a, b)).abs().max().item()
    gb = Bn * K * (M + N) * 2 / 1e9
    print(f"  {what:20s}: library {t1 * 1e3:7.1f} us ({gb / t1:5.2f} TB/s)   BLAS slabs {t0 * 1e3:7.1f} us ({gb / t0:5.2f} TB/s)   maxdiff {d:.3g}", flush=True)
