#!/bin/bash
# Round 4, GPU call 11: weight-gradient kernel with the next row fetched TWO steps before it is parked (SEGM_WGRAD_PD=2) - is the
# row loop bound by the fetch -> park latency?
mkdir -p gpurun_out
{
for v in 1 2 1 2; do echo "== SEGM_WGRAD_PD=$v"; SEGM_WGRAD_PD=$v timeout 300 python tools/gpu_conv_time.py 2>&1 | grep "^wgrad"; done
echo "== step"
for v in 1 2 1 2; do echo "SEGM_WGRAD_PD=$v"; SEGM_WGRAD_PD=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | grep -o "ms_per_step\": [0-9.]*"; done
} | tee gpurun_out/r04_wgrad_pd.log
SEGM_WGRAD_PD=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "wgrad or res_block or benchmarked" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_wgrad_pd.log
