#!/bin/bash
# Round 6, call 9: -exp(A_log) of the three directions as multi-tensor launches: Mamba parity on the GPU, step time
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "mamba or Mamba" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -2
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i: step ms', d['ms_per_step'], 'loss', d['config']['loss'])"
done 2>&1 | tee gpurun_out/r06_negexp3_step.log
