#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_dbg.out 2> gpurun_out/r05_bench_dbg.err
echo "rc=$?"
grep -v "MIOpen(HIP)\|GridwiseOp\|amdgpu.ids" gpurun_out/r05_bench_dbg.err | tail -40
tail -1 gpurun_out/r05_bench_dbg.out | cut -c1-1500
