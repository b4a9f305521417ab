#!/bin/bash
# Round 6, call 8: segm_wgrad_gemm TN (two chunks in flight, 48-column blocks, padded-row vector loads): parity, per-shape times, the step with it on / off
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad_gemm or wgemm" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -2
python tools/gpu_wgrad_gemm_time.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | head -17 | tee gpurun_out/r06_wgemm_tn_v3.log
for i in 1 2; do for f in 1 0; do
  SEGM_WGRAD_GEMM_TN=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_WGRAD_GEMM_TN=$f run $i: step ms', d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r06_wgemm_tn_step.log
