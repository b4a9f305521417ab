#!/bin/bash
# Round 2, second full GPU call: the whole -m gpu suite with the new defaults (projection kernel on, conv output / delta kept,
# new carry kernels), a bench line, the step profile.
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02_gpu_tests2.log 2>&1; tail -15 gpurun_out/r02_gpu_tests2.log
timeout 300 python bench.py > gpurun_out/r02_bench2.log 2> gpurun_out/r02_bench2.err; tail -1 gpurun_out/r02_bench2.log | cut -c1-900
bash tools/gpu_step_profile.sh r02_step
