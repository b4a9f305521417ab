#!/bin/bash
mkdir -p gpurun_out
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3d" > gpurun_out/conv_check_tests.log 2>&1
tail -3 gpurun_out/conv_check_tests.log
SEGM_CONV_VERBOSE=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/conv2_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen\|autotune" gpurun_out/conv2_bench.log | tail -1 | cut -c1-330
grep "autotune.*(2, 4, 128" gpurun_out/conv2_bench.log | cut -c1-200
