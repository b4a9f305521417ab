#!/bin/bash
# Round 3, GPU call 1: the new parity tests (bf16 whole network vs fp32, graph replay vs eager, inner-fn matrix, conv1d weight
# gradient tolerances), the bench line with the captured HIP-graph step next to the eager step, BASELINE config 1 / config 4.
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 900 python -m pytest tests/test_gpu_network_bf16.py -m gpu -q -x --durations=5 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_call1_tests.log; tail -25 gpurun_out/r03_call1_tests.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "causal_conv1d_reference_matrix" 2>&1 | tail -3
echo "== bench, graph"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_call1_bench_graph.json 2> gpurun_out/r03_call1_bench_graph.err; tail -c 3000 gpurun_out/r03_call1_bench_graph.json; tail -3 gpurun_out/r03_call1_bench_graph.err
echo "== bench, eager"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > gpurun_out/r03_call1_bench_eager.json 2> gpurun_out/r03_call1_bench_eager.err; cut -c1-600 gpurun_out/r03_call1_bench_eager.json; tail -3 gpurun_out/r03_call1_bench_eager.err
