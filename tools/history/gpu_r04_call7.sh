#!/bin/bash
# Round 4, GPU call 7: north star N1's last variant - dt_proj inside the forward scan passes (SEGM_SCAN_FUSED_DTPROJ=1):
# kernel-level parity + timings, step A/B, the network tests under the variant
mkdir -p gpurun_out
timeout 300 python tools/gpu_scan_fused_dtproj_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_scan_fused_dtproj.log
for v in 0 1 0 1; do echo "SEGM_SCAN_FUSED_DTPROJ=$v"; SEGM_SCAN_FUSED_DTPROJ=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-200; done 2>&1 | tee -a gpurun_out/r04_scan_fused_dtproj.log
SEGM_SCAN_FUSED_DTPROJ=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py tests/test_gpu_kernels.py -m gpu -q -x -k "not conv3" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_scan_fused_dtproj.log
