#!/bin/bash
# Round 4, GPU call 14: the chained 64-wide kernel with its row loop instantiated per K part (SEGM_CONV_CHAIN_VAR=11) against the
# shipped schedule (3)
mkdir -p gpurun_out
{
for v in 3 11 3 11; do echo "== SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids | sed 's/reduce [^)]*)  //'; done
echo "== step"
for v in 3 11 3 11; do echo "SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | grep -o "ms_per_step\": [0-9.]*"; done
} | tee gpurun_out/r04_conv_chain_perpart.log
SEGM_CONV_CHAIN_VAR=11 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "conv3 or conv_k3 or conv3d or res_block or benchmarked" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_conv_chain_perpart.log
