#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r6_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/r6_bench.log | tail -5
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r6_model_tests.log 2>&1
tail -5 gpurun_out/r6_model_tests.log
