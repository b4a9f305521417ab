#!/bin/bash
# Round 3, GPU call 5: backward main kernel with the state-major B / C tile (one LDS round trip per state instead of 32).
mkdir -p gpurun_out
timeout 600 python tools/gpu_scan_ab.py build/variants/r02.so build/variants/r03c.so build/variants/r03c_bwd1.so 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r03_scan_ab3.log
for c in 128 512; do echo "SEGM_CHUNK=$c"; SEGM_CHUNK=$c timeout 300 python tools/gpu_scan_ab.py build/variants/r03c.so 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/r03_scan_ab3.log; done
bash tools/gpu_scan_kernels.sh build/variants/r03c.so r03c 2>&1 | tail -14 | tee gpurun_out/r03_scan_kernels_v3.txt
