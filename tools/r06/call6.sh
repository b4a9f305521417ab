#!/bin/bash
# Round 6, call 6: peak memory of the training step, eager launches, SEGM_RECOMPUTE 0 / 1; then the GPU suite
mkdir -p gpurun_out
for rc in 0 1; do
SEGM_RECOMPUTE=$rc timeout 600 python bench.py --no-graph --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_RECOMPUTE=$rc eager: step ms', d['ms_per_step'], 'peak', {k: v for k, v in d['config']['peak_mem_mb'].items() if k.endswith('_mb')})"
done 2>&1 | tee gpurun_out/r06_peak_memory.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -8 | tee gpurun_out/r06_gpu_tests_call6.log
