#!/bin/bash
# Round 6: the channel-last 3x3x3 forward prototype (16-row tiles) against the shipped NCDHW kernel: dense rows and rows padded by 8 voxels
mkdir -p gpurun_out
for padx in 0 8 2; do
echo "== row pitch padded by $padx voxels"
SEGM_CL_PADX=$padx timeout 600 python tools/gpu_conv_cl_time.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -20
done | tee gpurun_out/r06_conv_cl_v4b.log
