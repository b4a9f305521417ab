#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "layernorm or segmamba or mamba" > gpurun_out/ln_check_tests.log 2>&1
tail -3 gpurun_out/ln_check_tests.log
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ln_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/ln_bench.log | tail -1 | cut -c1-330
