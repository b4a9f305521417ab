#!/bin/bash
# Round 4, GPU call 8: schedule variants of the chained 3x3x3 kernels (SEGM_CONV_CHAIN_VAR: bit 0 = K parts 1 - 3 skip their
# all-zero eleventh chunk, bits 1 - 2 = A fragments read 2 / 3 chunks ahead): per-shape timings, step time, conv parity on the
# variant that wins; then where the step time of the dt_proj-in-scan variant goes (kernel trace)
mkdir -p gpurun_out
for v in 0 1 2 3 4 5; do echo "== SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v SEGM_TIME_CONV_ONLY=1 timeout 200 python tools/gpu_chain_time.py 2>&1 | grep -v amdgpu.ids | sed 's/reduce [^)]*)  //'; done | tee gpurun_out/r04_conv_chain_var_time.log
echo "== step"
for v in 0 3 2 1 0 3; do echo "SEGM_CONV_CHAIN_VAR=$v"; SEGM_CONV_CHAIN_VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-220; done | tee gpurun_out/r04_conv_chain_var_step.log
echo "== conv parity under variant 3"
SEGM_CONV_CHAIN_VAR=3 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "conv3 or conv_k3 or conv3d or res_block or benchmarked" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -4 | tee gpurun_out/r04_call8_parity.log
echo "== dt_proj in the scan: kernel trace of the step"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SEGM_SCAN_FUSED_DTPROJ=$v timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_dt$v -o dt$v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-configs --no-roofline --no-graph > /dev/null 2>&1
  f=$(find /tmp/prof_dt$v -name "*kernel_stats.csv" | head -1)
  echo "SEGM_SCAN_FUSED_DTPROJ=$v"; grep -i "scan_fwd\|linear_rows\|scan_carry" "$f" | cut -c1-200
done | tee $GRAFT_REPO_ROOT/gpurun_out/r04_dtproj_step_kernels.log
