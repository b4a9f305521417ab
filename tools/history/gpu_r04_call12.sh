#!/bin/bash
# Round 4, GPU call 12: weight-gradient kernel as six waves (ky x half of the x block; two workgroups = 3 waves on every SIMD)
# against the three-wave layout (2, 2, 1, 1): SEGM_WGRAD_SIX=1
mkdir -p gpurun_out
{
for v in 0 1 0 1; do echo "== SEGM_WGRAD_SIX=$v"; SEGM_WGRAD_SIX=$v timeout 300 python tools/gpu_conv_time.py 2>&1 | grep "^wgrad"; done
echo "== step"
for v in 0 1 0 1; do echo "SEGM_WGRAD_SIX=$v"; SEGM_WGRAD_SIX=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | grep -o "ms_per_step\": [0-9.]*"; done
} | tee gpurun_out/r04_wgrad_six.log
SEGM_WGRAD_SIX=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "wgrad or res_block or benchmarked" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_wgrad_six.log
