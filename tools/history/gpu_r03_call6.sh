#!/bin/bash
# Round 3, GPU call 6: both backward main kernels in one library (SEGM_BWD_MAIN=r2 | r3; r3 with A[d][n] prefetched a state ahead),
# then the whole step with the three-direction launches.
mkdir -p gpurun_out
for m in r2 r3; do echo "SEGM_BWD_MAIN=$m"; SEGM_BWD_MAIN=$m timeout 300 python tools/gpu_scan_ab.py build/variants/r03d.so 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/r03_scan_ab4.log; done
echo "== bench (fused three-direction node, table routing, graph)"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline > gpurun_out/r03_call6_bench.json 2> gpurun_out/r03_call6_bench.err; cut -c1-200 gpurun_out/r03_call6_bench.json; grep -v "MIOpen\|Gridwise" gpurun_out/r03_call6_bench.err | tail -3
echo "== bench SEGM_MAMBA_FUSED3=0"
SEGM_MAMBA_FUSED3=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-200
echo "== bench eager"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200
echo "== step profile"
bash tools/gpu_step_profile.sh r03_step_v1 2>&1 | tail -45
