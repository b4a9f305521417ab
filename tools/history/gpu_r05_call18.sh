#!/bin/bash
# Round 5, call 18: wgemm_tn with ds_read_b64_tr_b16 fragments: GPU parity of the kernel, then the step with SEGM_WGRAD_GEMM_TN=1 / 0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad_gemm or wgemm" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -3 | tee gpurun_out/r05_wgemm_tr_tests.log
for i in 1 2; do for f in 1 0; do
  SEGM_WGRAD_GEMM_TN=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_WGRAD_GEMM_TN=$f run $i: step ms', d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_wgemm_tr_step.log
