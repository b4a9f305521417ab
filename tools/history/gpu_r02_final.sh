#!/bin/bash
# Round 2, final GPU call: the whole -m gpu suite, the bench line, the step profile.  (The HBM-side traffic of the scan was
# measured at commit 5ebb8d5 by tools/gpu_pmc_traffic.sh; the scan kernels have not changed since.)
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 420 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r02_gpu_tests_final.log; tail -14 gpurun_out/r02_gpu_tests_final.log
timeout 200 python bench.py > gpurun_out/r02_bench_final.log 2> gpurun_out/r02_bench_final.err; tail -1 gpurun_out/r02_bench_final.log | cut -c1-1500
bash tools/gpu_step_profile.sh r02_step_final2 2>&1 | tail -12
