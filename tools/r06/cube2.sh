#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cube" > gpurun_out/r06_cube_tests.log 2>&1; tail -3 gpurun_out/r06_cube_tests.log
timeout 900 python -m pytest tests/test_gpu_blocks_conditioned.py tests/test_gpu_model.py -q -m gpu -x > gpurun_out/r06_cube_tests2.log 2>&1; tail -3 gpurun_out/r06_cube_tests2.log
bash tools/gpu_step_profile.sh r06_cube_step > /dev/null 2>&1; head -12 gpurun_out/r06_cube_step_step_kernels.txt; grep -n "scatter_gather\|cube" gpurun_out/r06_cube_step_step_kernels.txt
