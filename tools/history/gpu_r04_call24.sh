#!/bin/bash
# Round 4, GPU call 24: space-to-depth kernel in the kernel-2 stride-2 down-sampling convolutions: step A/B (SEGM_D2S_HIP), network parity
mkdir -p gpurun_out
{
for v in 1 0 1 0; do echo "SEGM_D2S_HIP=$v"; SEGM_D2S_HIP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; done
} | tee gpurun_out/r04_s2d_down.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "segmamba or network" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_s2d_down.log
