#!/bin/bash
# Round 2, first GPU call: VALU probe with in-kernel clock, the whole -m gpu suite at the north-star tolerances (incl. the
# at-size fp64-oracle tests), the projection kernel against BLAS, a bench line.
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
timeout 120 ./build/probe_valu2 > gpurun_out/r02_probe_valu2.log 2>&1; tail -50 gpurun_out/r02_probe_valu2.log
timeout 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r02_gpu_tests.log 2>&1; tail -40 gpurun_out/r02_gpu_tests.log
timeout 90 python - > gpurun_out/r02_linear.log 2>&1 <<'PY'
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
cat gpurun_out/r02_linear.log
timeout 240 python bench.py > gpurun_out/r02_bench1.log 2> gpurun_out/r02_bench1.err; tail -1 gpurun_out/r02_bench1.log | cut -c1-1500
