#!/bin/bash
# Round 4, GPU call 21: segm_linear_rows accumulate mode with the old outputs prefetched: stand-alone timings, step
mkdir -p gpurun_out
{
timeout 300 python tools/gpu_linear_acc_time.py 2>&1 | grep -v amdgpu.ids
echo "== step"
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; done
} | tee gpurun_out/r04_linear_acc.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "linear or mamba or network or inner" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_linear_acc.log
