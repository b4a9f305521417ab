#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "transpose or token_layout" > gpurun_out/r8_tests.log 2>&1
tail -3 gpurun_out/r8_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r8_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/r8_bench.log | tail -3 | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r8_model_tests.log 2>&1
tail -3 gpurun_out/r8_model_tests.log
