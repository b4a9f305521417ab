#!/bin/bash
# Round 3, GPU call 8: SEGM_CONV_CAT_FUSED A/B (written in round 2, never timed), shapes of the remaining ATen copies / adds / sums,
# wgrad partial-sum kernel with eight loads in flight.
mkdir -p gpurun_out
for v in 0 1; do echo "== SEGM_CONV_CAT_FUSED=$v"; SEGM_CONV_CAT_FUSED=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-200; done
echo "== model tests with SEGM_CONV_CAT_FUSED=1"
SEGM_CONV_CAT_FUSED=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -4
echo "== op shapes"
SEGM_CONV_CAT_FUSED=1 timeout 600 python tools/gpu_copy_shapes.py 2>&1 | grep -v "MIOpen\|Gridwise" | tee gpurun_out/r03_copy_shapes.log | head -90
echo "== step profile (cat fused)"
SEGM_CONV_CAT_FUSED=1 bash tools/gpu_step_profile.sh r03_step_v3 2>&1 | tail -30
