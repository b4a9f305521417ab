#!/bin/bash
# Round 4, GPU call 17: 32-wide chained kernel with fragments two chunks ahead (SEGM_CONV_CHAIN32_PF2=1); step with the routing
# table sending 64^3 layers to the 32-wide kernel
mkdir -p gpurun_out
{
for v in 0 1 0 1; do echo "== SEGM_CONV_CHAIN32_PF2=$v"; SEGM_CONV_CHAIN32_PF2=$v SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids | sed 's/reduce [^)]*)  //; s/  chain [^)]*)//'; done
echo "== step"
for v in 0 1 0 1; do echo "SEGM_CONV_CHAIN32_PF2=$v"; SEGM_CONV_CHAIN32_PF2=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; done
} | tee gpurun_out/r04_conv_chain32_pf2.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "conv3 or conv_k3 or conv3d or res_block or benchmarked or network" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_conv_chain32_pf2.log
