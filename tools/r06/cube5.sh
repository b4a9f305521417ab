#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cube" > gpurun_out/r06_cube_tests.log 2>&1; grep -v "GridwiseOp\|MIOpen(HIP)" gpurun_out/r06_cube_tests.log | tail -2
timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r06_bench_cube4.err | tee gpurun_out/r06_bench_cube4.json | cut -c1-330
SEGM_CONV_CUBE_WGRAD_MAX_WIDTH=8 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-330
