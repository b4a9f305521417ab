#!/bin/bash
# Round 6, call 4: cache policy of the write-once / read-once streams of the scan (checkpoints, un-gated y), three rounds
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 900 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so build/variants/libsegm_nt_a_ck.so build/variants/libsegm_nt_b_ck_out.so build/variants/libsegm_nt_c_ck_out_bwd.so 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)\|float32"
done | tee gpurun_out/r06_scan_nt_policy.log
