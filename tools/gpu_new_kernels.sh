#!/bin/bash
# GPU look at the conv forward kernels (row staging, chained K parts) and the training-step kernels: parity tests, A/B timings
mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d" > gpurun_out/new_kernels_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/new_kernels_tests.log
tail -3 gpurun_out/new_kernels_tests.log
timeout 45 python tools/gpu_chain_time.py > gpurun_out/new_kernels_time.log 2>&1
echo "time rc=$?" >> gpurun_out/new_kernels_time.log
grep -v amdgpu.ids gpurun_out/new_kernels_time.log
