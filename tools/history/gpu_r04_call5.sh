#!/bin/bash
# Round 4, GPU call 5: the conditioned per-block tests with the floor-relative criterion, the fp16 network test, scan repeatability
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_blocks_conditioned.py tests/test_gpu_kernels.py -m gpu -q -rf -k "conditioned or fp16 or repeatability or benchmarked_shape" > gpurun_out/r04_call5_full.log 2>&1
grep -v "GridwiseOp\|amdgpu.ids" gpurun_out/r04_call5_full.log | tail -60 | cut -c1-300 | tee gpurun_out/r04_gpu_tests_call5.log
