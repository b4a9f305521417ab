#!/bin/bash
# the whole -m gpu suite on the closing commit of round 3
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 1700 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_gpu_tests_closing.log; tail -12 gpurun_out/r03_gpu_tests_closing.log
