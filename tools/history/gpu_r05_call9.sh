#!/bin/bash
# Round 5, call 9: statistics epilogue - parity tests, step A/B (SEGM_CONV_STATS 0 / 1, with and without SEGM_CONV_CAT_FUSED), step profile
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "statistics or conv or block or segmamba or network or instnorm or norm" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -6 | tee gpurun_out/r05_call9_tests.log
for v in "1 0" "0 0" "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $v
  SEGM_CONV_STATS=$1 SEGM_CONV_CAT_FUSED=$2 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_CONV_STATS=$1 SEGM_CONV_CAT_FUSED=$2', d['ms_per_step'], d['config']['loss'])" | tee -a gpurun_out/r05_inorm_epilogue_step.log
done
bash tools/gpu_step_profile.sh r05_call9 2>&1 | tail -36
