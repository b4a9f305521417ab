#!/bin/bash
# Round 3, closing GPU call (after tools/gpu_r03_final.sh: 350 passed, 1 tolerance case): the at-size tests again, the tests of the kernel
# added since (segm_channel_sum) and of the model, the default bench line, the per-kernel table of one step.
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 1200 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_model.py tests/test_gpu_network_bf16.py tests/test_gpu_kernels.py -m gpu -q -k "config4 or channel_sum or skinny or model or network or segmamba or training" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_gpu_tests_final2.log; tail -4 gpurun_out/r03_gpu_tests_final2.log
timeout 900 python bench.py > gpurun_out/r03_bench_final2.json 2> gpurun_out/r03_bench_final2.err; echo "bench rc=$?"; tail -1 gpurun_out/r03_bench_final2.json | cut -c1-400
bash tools/gpu_step_profile.sh r03_step_final2 2>&1 | tail -12
