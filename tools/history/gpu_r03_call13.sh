#!/bin/bash
# Round 3, GPU call 13: ablations of the forward apply kernel in its throughput-bound regime (6 waves per SIMD of work, 3 resident... 5 with 94 registers)
mkdir -p gpurun_out
for n in base nostore nogate nockpt nolds noload noy; do
  echo "== $n"
  SEGM_OCC_LIB=$GRAFT_REPO_ROOT/build/variants/abl_$n.so SEGM_OCC_K=3,6 bash tools/gpu_scan_occupancy.sh abl_$n 2>&1 | grep -E "apply|fwd_agg|waves/SIMD [36]:"
done 2>&1 | tee gpurun_out/r03_scan_ablations.log
