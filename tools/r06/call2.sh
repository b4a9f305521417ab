#!/bin/bash
# Round 6, call 2: forward scan timeline with / without the checkpoint stores (bf16, one and three directions)
mkdir -p gpurun_out
TIMELINE_NOCKPT=1 TIMELINE_ONLY=bf16-1 timeout 600 python tools/gpu_scan_timeline.py build/variants/libsegm_timeline.so gpurun_out/r06_scan_timeline_nockpt_1.txt 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | grep "===\|cycles per wave-step\|launch span\|steady sub"
TIMELINE_NOCKPT=1 TIMELINE_ONLY=bf16-3 timeout 600 python tools/gpu_scan_timeline.py build/variants/libsegm_timeline.so gpurun_out/r06_scan_timeline_nockpt_3.txt 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | grep "===\|cycles per wave-step\|launch span\|steady sub"
