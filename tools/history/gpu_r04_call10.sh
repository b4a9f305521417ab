#!/bin/bash
# Round 4, GPU call 10: the data-parallel path over RCCL on one rank (hooks, side stream, sliced all-reduce; graph + one call),
# cat(up, skip) convolutions as one node with in-place accumulation (SEGM_CONV_CAT_FUSED=1) re-timed, weight-gradient sums after
# the reduce4 rule
mkdir -p gpurun_out
echo "== ddp tests"
timeout 1500 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -15 | tee gpurun_out/r04_gpu_ddp_tests.log
echo "== cat fused"
for v in 0 1 0 1; do echo "SEGM_CONV_CAT_FUSED=$v"; SEGM_CONV_CAT_FUSED=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | grep -o "ms_per_step\": [0-9.]*"; done | tee gpurun_out/r04_cat_fused_step.log
SEGM_CONV_CAT_FUSED=1 timeout 900 python -m pytest tests/test_gpu_network_bf16.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_cat_fused_step.log
