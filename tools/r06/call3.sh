#!/bin/bash
# Round 6, call 3: checkpoint store forms of the forward apply pass (timing experiment; the backward still reads pairs)
mkdir -p gpurun_out
for i in 1 2; do
timeout 900 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so build/variants/libsegm_ck_nt.so build/variants/libsegm_ck_quad.so build/variants/libsegm_ck_quadnt.so 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)"
done | tee gpurun_out/r06_scan_ckpt_forms.log
