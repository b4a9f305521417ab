#!/bin/bash
# Round 4, GPU call 1: the 8-step-window backward main kernel (scan_bwd_w8.hip) on the hardware - parity, then time:
# three waves per SIMD with assembly DPP stages (the default) against two waves per SIMD and against the intrinsic stages.
mkdir -p gpurun_out
echo "== scan parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -5 | tee gpurun_out/r04_call1_parity.log
echo "== variants"
timeout 600 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so build/variants/r04_w2.so build/variants/r04_noasm.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_scan_ab1.log
echo "== kernels"
bash tools/gpu_scan_kernels.sh segmamba_amd/libsegmamba_hip.so r04a 2>&1 | tee gpurun_out/r04_scan_kernels_a.txt
echo "== model parity + step"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r04_call1_model.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r04_bench_call1.json
