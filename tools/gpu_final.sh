#!/bin/bash
# model-level GPU tests, then the default bench line
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/final_model_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/final_model_tests.log
tail -4 gpurun_out/final_model_tests.log
timeout 100 python bench.py > gpurun_out/final_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/final_bench.log
grep -v amdgpu.ids gpurun_out/final_bench.log | tail -5
