#!/bin/bash
# Round 5, call 17: dB / dC vector stores + batched partial reductions: scan / mamba parity tests, step time (3 runs)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "scan or mamba or conv1d or dbc or selective" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -3 | tee gpurun_out/r05_call17_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i: step ms', d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05_call17_step.log
