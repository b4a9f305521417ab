#!/bin/bash
mkdir -p gpurun_out
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3d" > gpurun_out/r13_conv_tests.log 2>&1
tail -4 gpurun_out/r13_conv_tests.log
timeout 300 python tools/gpu_wgrad_time.py > gpurun_out/r13_conv_time.log 2>&1
grep -v amdgpu.ids gpurun_out/r13_conv_time.log | tail -8
SEGM_CONV_VERBOSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r13_bench.log 2>&1
grep -v "amdgpu.ids\|MIOpen\|autotune" gpurun_out/r13_bench.log | tail -3 | cut -c1-300
grep "'fwd'\|'dgrad'" gpurun_out/r13_bench.log | cut -c1-200 | head -30
