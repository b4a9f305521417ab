#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "instance_norm" > gpurun_out/r7_norm_tests.log 2>&1
tail -5 gpurun_out/r7_norm_tests.log
timeout 300 python tools/gpu_norm_time.py > gpurun_out/r7_norm_time.log 2>&1
grep -v amdgpu.ids gpurun_out/r7_norm_time.log | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r7_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/r7_bench.log | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r7_model_tests.log 2>&1
tail -5 gpurun_out/r7_model_tests.log
