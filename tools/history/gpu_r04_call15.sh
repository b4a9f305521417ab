#!/bin/bash
# Round 4, GPU call 15: chained kernels with per-part loops + unmasked park (64-wide VAR 11, 32-wide VAR 9) - timings, parity, step
mkdir -p gpurun_out
{
for v in 3 0 3; do echo "== SEGM_CONV_CHAIN_VAR=$v (3 = shipped set, 0 = round-3 schedule)"; SEGM_CONV_CHAIN_VAR=$v SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids | sed 's/reduce [^)]*)  //'; done
echo "== step"
for v in 3 0 3; do echo "SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; done
} | tee gpurun_out/r04_conv_chain_final.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "conv3 or conv_k3 or conv3d or res_block or benchmarked or network" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_conv_chain_final.log
