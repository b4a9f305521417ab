#!/bin/bash
# Round 3, GPU call 10: skinny TN v2 (16-byte loads) against the other routes; forward scan kernels with compile-time step flags,
# four waves per SIMD in the apply kernel, chunk 128; step time.
mkdir -p gpurun_out
echo "== skinny_tn timing"
timeout 300 python tools/gpu_skinny_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_skinny_tn_time.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "skinny" 2>&1 | tail -2
echo "== scan A/B"
timeout 900 python tools/gpu_scan_ab.py SEGM_SCAN_FLAGS=0:build/variants/r03f.so build/variants/r03f.so build/variants/r03f_w4.so SEGM_CHUNK=128:build/variants/r03f.so SEGM_CHUNK=128:build/variants/r03f_w4.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_scan_ab5.log
echo "== scan parity (fast kernels)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -3
echo "== bench"
for v in 0 1; do echo "SEGM_SKINNY_TN=$v"; SEGM_SKINNY_TN=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200; done
