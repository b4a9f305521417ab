#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x > gpurun_out/r06_cube_tests3.log 2>&1; grep -v "GridwiseOp\|MIOpen(HIP)" gpurun_out/r06_cube_tests3.log | tail -3
timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r06_bench_cube5.err | tee gpurun_out/r06_bench_cube5.json | cut -c1-330
bash tools/gpu_step_profile.sh r06_cube5_step > /dev/null 2>&1; head -8 gpurun_out/r06_cube5_step_step_kernels.txt; grep -n "scatter_gather\|gather16\|cube" gpurun_out/r06_cube5_step_step_kernels.txt
