#!/bin/bash
mkdir -p gpurun_out
SEGM_CONV_VERBOSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_bench.log | tail -40
