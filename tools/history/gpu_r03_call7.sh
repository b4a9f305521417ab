#!/bin/bash
# Round 3, GPU call 7: the step with one-gather weight packs, gathered gradients, direction-batched conv1d; model-level parity.
mkdir -p gpurun_out
echo "== bench graph"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline > gpurun_out/r03_call7_bench.json 2> gpurun_out/r03_call7_bench.err; cut -c1-200 gpurun_out/r03_call7_bench.json; grep -v "MIOpen\|Gridwise" gpurun_out/r03_call7_bench.err | tail -3
echo "== bench eager"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200
echo "== model tests"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -6
echo "== step profile"
bash tools/gpu_step_profile.sh r03_step_v2 2>&1 | tail -32
