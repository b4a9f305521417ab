#!/bin/bash
# Round 3, GPU call 4: backward main kernel with the dB / dC tile + two flushes per window (call 3: scattered per-state atomics
# made it 3 x slower than round 2), against round 2's library.
mkdir -p gpurun_out
timeout 600 python tools/gpu_scan_ab.py build/variants/r02.so build/variants/r03b.so build/variants/r03b_bwd1.so 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r03_scan_ab2.log
bash tools/gpu_scan_kernels.sh build/variants/r03b.so r03b 2>&1 | tail -14 | tee gpurun_out/r03_scan_kernels_v2.txt
