#!/bin/bash
# Round 4, GPU call 6: the chained 3x3x3 kernel with the two waves of a SIMD in opposite phases (conv3d_k3_fwd48_pp_kernel)
# against the lock-step kernel: timings per layer shape, parity of the convolution tests, step time
mkdir -p gpurun_out
for pp in 1 0; do echo "== SEGM_CONV_PP=$pp"; SEGM_CONV_PP=$pp SEGM_TIME_CONV_ONLY=1 timeout 300 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r04_conv_pp_time.log
echo "== conv parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "conv3 or conv_k3 or conv3d or res_block or benchmarked" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -4 | tee gpurun_out/r04_call6_parity.log
echo "== step"
for pp in 1 0; do echo "SEGM_CONV_PP=$pp"; SEGM_CONV_PP=$pp timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-220; done | tee gpurun_out/r04_conv_pp_step.log
