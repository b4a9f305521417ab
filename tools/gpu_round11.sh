#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan" > gpurun_out/r11_scan_tests.log 2>&1
tail -3 gpurun_out/r11_scan_tests.log
timeout 600 python tools/gpu_scan_sweep.py > gpurun_out/r11_scan_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/r11_scan_sweep.log
