#!/bin/bash
# Round 3, GPU call 16: the scan with the causal conv1d inside its launches (north star clause N1, opt-in): parity at the stage-0 size,
# time against the separate launch, and the training step with it switched on.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv1d_inside or scan_segmamba_shapes" 2>&1 | tail -3
timeout 300 python tools/gpu_scan_fused_conv_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_scan_fused_conv.log
for v in 0 1; do echo "SEGM_SCAN_FUSED_CONV1D=$v"; SEGM_SCAN_FUSED_CONV1D=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200; done 2>&1 | tee -a gpurun_out/r03_scan_fused_conv.log
SEGM_SCAN_FUSED_CONV1D=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -3
