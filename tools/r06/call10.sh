#!/bin/bash
# Round 6, call 10: linear_rows for K > 192 and for a few thousand rows (the projections of stages 2 / 3): parity, the step with the row threshold at 1024 / 32768
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "linear or mamba or Mamba or inner or segmamba or network or SegMamba" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -3
for i in 1 2 3; do for m in 1024 32768; do
  SEGM_LINEAR_ROWS_MIN=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_LINEAR_ROWS_MIN=$m run $i: step ms', d['ms_per_step'], 'loss', d['config']['loss'])"
done; done 2>&1 | tee gpurun_out/r06_linear_k_step.log
