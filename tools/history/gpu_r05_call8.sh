#!/bin/bash
# Round 5, call 8: forward chained kernels after the static-ky slot arithmetic, the statistics-epilogue probe, wgrad 32-wide / 3 per CU,
# cat-fused A/B on the step, step profile
mkdir -p gpurun_out
SEGM_TIME_CONV_ONLY=1 timeout 600 python tools/gpu_chain_time.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tee gpurun_out/r05_conv_chain_static_ky.log
timeout 600 python tools/history/r05/inorm_epilogue.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tee gpurun_out/r05_inorm_epilogue.log
timeout 600 python tools/history/r05/wgrad_ab.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tee gpurun_out/r05_wgrad_ab3.log
for v in 0 1 0 1; do
  SEGM_CONV_CAT_FUSED=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_CONV_CAT_FUSED=$v', d['ms_per_step'])" | tee -a gpurun_out/r05_cat_fused_step.log
done
bash tools/gpu_step_profile.sh r05_call8 2>&1 | tail -40
