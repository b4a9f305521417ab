#!/bin/bash
# Round 4, GPU call 9: (a) static priority for the younger half of the chained 64-wide kernel (SEGM_CONV_CHAIN_VAR=11) against the
# shipped schedule (3); (b) weight-gradient partial sums with 16-byte loads (SEGM_WGRAD_REDUCE4=1); (c) what ATen / BLAS still
# run per step (shapes), for the vendor-remnant list
mkdir -p gpurun_out
{
for v in 3 11 3 11; do echo "== SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids | sed 's/reduce [^)]*)  //' | head -1; done
for v in 0 1 0 1; do echo "== SEGM_WGRAD_REDUCE4=$v"; SEGM_WGRAD_REDUCE4=$v timeout 300 python tools/gpu_conv_time.py 2>&1 | grep "^wgrad"; done
echo "== step"
for e in "SEGM_CONV_CHAIN_VAR=3 SEGM_WGRAD_REDUCE4=0" "SEGM_CONV_CHAIN_VAR=11 SEGM_WGRAD_REDUCE4=0" "SEGM_CONV_CHAIN_VAR=3 SEGM_WGRAD_REDUCE4=1" "SEGM_CONV_CHAIN_VAR=3 SEGM_WGRAD_REDUCE4=0" "SEGM_CONV_CHAIN_VAR=3 SEGM_WGRAD_REDUCE4=1"; do echo "$e"; env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | grep -o "ms_per_step\": [0-9.]*"; done
} | tee gpurun_out/r04_call9_ab.log
echo "== op shapes"
timeout 400 python tools/gpu_copy_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | cut -c1-200 > gpurun_out/r04_copy_shapes.log; head -60 gpurun_out/r04_copy_shapes.log
