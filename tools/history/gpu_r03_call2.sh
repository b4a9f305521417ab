#!/bin/bash
# Round 3, GPU call 2: bench with the captured step (call 1 lost the line to a config-4 error), per-layer bf16 error table,
# the rest of the new tests.
mkdir -p gpurun_out
echo "== bench, graph"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_call2_bench_graph.json 2> gpurun_out/r03_call2_bench_graph.err; tail -c 4500 gpurun_out/r03_call2_bench_graph.json; grep -v "MIOpen\|GridwiseOp" gpurun_out/r03_call2_bench_graph.err | tail -5
echo "== layer errors"
timeout 600 python tools/gpu_bf16_layer_errors.py 64 2>&1 | grep -v "MIOpen\|GridwiseOp" > gpurun_out/r03_bf16_layer_errors.log; tail -150 gpurun_out/r03_bf16_layer_errors.log
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_network_bf16.py -m gpu -q --durations=5 --deselect tests/test_gpu_network_bf16.py::test_segmamba_bf16_library_path_matches_fp32_fwd_bwd_64cube 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_call2_tests.log; tail -25 gpurun_out/r03_call2_tests.log
