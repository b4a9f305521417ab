#!/bin/bash
# Round 6: the channel-last 3x3x3 forward prototype against the shipped NCDHW kernel, input prefetch depth 2 and 6 k-steps
mkdir -p gpurun_out
for pf in 2 6; do
echo "== SEGM_CL_PREFETCH=$pf"
SEGM_CL_PREFETCH=$pf timeout 600 python tools/gpu_conv_cl_time.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -20
done | tee gpurun_out/r06_conv_cl_v3.log
