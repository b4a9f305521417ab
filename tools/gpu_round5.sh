#!/bin/bash
# session 5: wgrad kernel parity + timing, scan sweep, bench
mkdir -p gpurun_out
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3d" > gpurun_out/r5_wgrad_tests.log 2>&1
tail -5 gpurun_out/r5_wgrad_tests.log
timeout 300 python tools/gpu_wgrad_time.py > gpurun_out/r5_wgrad_time.log 2>&1
cat gpurun_out/r5_wgrad_time.log | tail -8
timeout 400 python tools/gpu_scan_sweep.py > gpurun_out/r5_scan_sweep.log 2>&1
cat gpurun_out/r5_scan_sweep.log | tail -30
SEGM_CONV_VERBOSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_bench.log 2>&1
tail -12 gpurun_out/r5_bench.log
