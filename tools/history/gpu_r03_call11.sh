#!/bin/bash
# Round 3, GPU call 11: forward scan kernels with ring prefetch of the row streams (registers 134 -> 94, no VMEM drain at the loop
# head) against the previous build; skinny TN with eight passes in flight.
mkdir -p gpurun_out
echo "== scan A/B"
timeout 900 python tools/gpu_scan_ab.py build/variants/r03f.so build/variants/r03g.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_scan_ab6.log
echo "== scan parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -3
echo "== skinny_tn timing"
timeout 300 python tools/gpu_skinny_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_skinny_tn_time2.log
echo "== per-kernel"
bash tools/gpu_scan_kernels.sh build/variants/r03g.so r03g 2>&1 | tee gpurun_out/r03_scan_kernels_v4.txt | tail -16
