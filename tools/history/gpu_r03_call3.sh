#!/bin/bash
# Round 3, GPU call 3: the rewritten regular-shape scan kernels (buffer addressing, LDS-tile prefetch, two waves per SIMD in the
# backward main kernel) against round 2's library on the roofline shape, their parity tests, per-kernel durations.
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
echo "== A/B"
timeout 600 python tools/gpu_scan_ab.py build/variants/r02.so build/variants/r03.so build/variants/r03_bwd1.so build/variants/r03_fwd4.so 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r03_scan_ab1.log
echo "== scan parity tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan or size or stress or config" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -15 | tee gpurun_out/r03_call3_scan_tests.log
echo "== per-kernel"
bash tools/gpu_scan_kernels.sh build/variants/r03.so r03 2>&1 | tail -14 | tee gpurun_out/r03_scan_kernels_v1.txt
echo "== model tests"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -15 | tee gpurun_out/r03_call3_model_tests.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/r03_call3_bench.json 2> gpurun_out/r03_call3_bench.err; cut -c1-250 gpurun_out/r03_call3_bench.json; grep -o '"roofline": {[^}]*}' gpurun_out/r03_call3_bench.json | cut -c1-400; grep -o '"backward": {[^}]*}' gpurun_out/r03_call3_bench.json | head -2
