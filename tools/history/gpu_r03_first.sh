#!/bin/bash
# First GPU call of the next round (the whole -m gpu suite needs ~7 min on a fresh box; the first `import torch` 1 - 2 min):
#   1. the full -m gpu suite (round 2 ended with the suite verified in pieces: profiles/r02_gpu_tests_final2_summary.log);
#   2. bench A/B of the switches that were written after the GPU budget of round 2 was spent (SEGM_CONV_CAT_FUSED) - flip the
#      default in segmamba_amd/conv3d.py if it wins and the suite is green with it;
#   3. the step profile for profiles/.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r03_first.sh'
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 900 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_gpu_tests_first.log; tail -14 gpurun_out/r03_gpu_tests_first.log
for v in 0 1; do
  echo "== SEGM_CONV_CAT_FUSED=$v"
  SEGM_CONV_CAT_FUSED=$v timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-260
done
SEGM_CONV_CAT_FUSED=1 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -3
bash tools/gpu_step_profile.sh r03_step_first 2>&1 | tail -12
