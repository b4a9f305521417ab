#!/bin/bash
mkdir -p gpurun_out
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3d" > gpurun_out/conv_check_tests.log 2>&1
tail -3 gpurun_out/conv_check_tests.log
timeout 300 python tools/gpu_conv_time.py > gpurun_out/conv_check_time.log 2>&1
grep "^fwd" gpurun_out/conv_check_time.log
SEGM_CONV_FWD_KZ_SPLIT=1 timeout 300 python tools/gpu_conv_time.py 2>&1 | grep "^fwd"
