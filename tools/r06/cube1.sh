#!/bin/bash
# cube kernel: GPU parity tests, the dispatcher's at-size tests, per-layer table through the routing, one bench run
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cube" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_blocks_conditioned.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python tools/gpu_conv_layers.py gpurun_out/r06_conv_layers_cube.txt 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tail -22
timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r06_bench_cube1.err | tee gpurun_out/r06_bench_cube1.json | cut -c1-400
SEGM_CONV_CUBE=0 timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-200
