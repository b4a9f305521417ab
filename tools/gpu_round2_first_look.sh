#!/bin/bash
# First GPU call of the next round: everything that was written after round 1's GPU budget ran out, in the order that matters.
#   1. parity of the new kernels (conv variants, decode, projection kernel)      -> gpurun_out/r2_new_tests.log
#   2. A/B timings of every forward-convolution variant + train-step glue        -> gpurun_out/r2_ab.log
#   3. projection kernel vs BLAS on the Mamba block's GEMM shapes                 -> gpurun_out/r2_linear.log
#   4. bench line with the dispatcher's choices printed                           -> gpurun_out/r2_bench.log
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "conv3d or decode or prefill or linear_rows or train_step" > gpurun_out/r2_new_tests.log 2>&1
tail -3 gpurun_out/r2_new_tests.log
timeout 60 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_ab.log
cat gpurun_out/r2_ab.log
timeout 60 python - > gpurun_out/r2_linear.log 2>&1 <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for M, K, N in [(524288, 48, 192), (524288, 96, 48), (524288, 96, 40), (524288, 40, 96), (524288, 8, 96), (524288, 96, 4),
                (65536, 96, 384), (65536, 192, 96), (65536, 192, 40)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (0.1 * torch.randn(N, K, device="cuda")).bfloat16()
    t_blas = timeit(lambda: torch.nn.functional.linear(x, w))
    t_hip = timeit(lambda: ops_raw.linear_rows(hip, x, w))
    gb = (M * K + M * N) * 2 / 1e9
    print(f"linear {M}x{K} -> {N}: BLAS {t_blas*1e3:.0f} us ({gb/t_blas*1e3:.0f} GB/s)  segm_linear_rows {t_hip*1e3:.0f} us ({gb/t_hip*1e3:.0f} GB/s)", flush=True)
PY
cat gpurun_out/r2_linear.log
SEGM_CONV_FWD_UNTIMED=1 SEGM_CONV_VERBOSE=1 timeout 150 python bench.py 2>&1 | grep -v "amdgpu.ids\|MIOpen" > gpurun_out/r2_bench.log
grep -c autotune gpurun_out/r2_bench.log; tail -1 gpurun_out/r2_bench.log | cut -c1-400
