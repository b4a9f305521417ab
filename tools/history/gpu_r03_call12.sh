#!/bin/bash
# Round 3, GPU call 12: (1) how the scan kernels' time scales with the waves a SIMD runs (throughput- or latency-bound?),
# (2) the backward main kernel on half windows at three waves per SIMD against the 16-step-window kernel.
mkdir -p gpurun_out
echo "== occupancy scaling"
bash tools/gpu_scan_occupancy.sh r03h 2>&1 | tee gpurun_out/r03_scan_occupancy.log
echo "== backward main kernels"
timeout 900 python tools/gpu_scan_ab.py build/variants/r03h.so SEGM_BWD_MAIN=half:build/variants/r03h.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_scan_ab7.log
echo "== parity with the half-window kernel"
SEGM_BWD_MAIN=half timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -3
