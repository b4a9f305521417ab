#!/bin/bash
# Round 5, call 15: the chained convolution kernels' 16-byte epilogue (SEGM_CONV_WIDE, per launch): stand-alone shapes, the kernel /
# block / network parity tests, then the step with / without, interleaved
mkdir -p gpurun_out
for f in 0 1 0 1; do echo "== SEGM_CONV_WIDE=$f"; SEGM_CONV_WIDE=$f SEGM_TIME_CONV_ONLY=1 python tools/gpu_chain_time.py 2>&1 | grep "conv fwd" | sed 's/reduce.*chain48/chain48/'; done | tee gpurun_out/r05_conv_wide.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "conv or block or segmamba or graphed or res_front or up_block or gsc or statistics" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -4 | tee gpurun_out/r05_conv_wide_tests.log
for i in 1 2 3; do
  for f in 1 0; do
    SEGM_CONV_WIDE=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/cw_${f}_${i}.json
    python - $f $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/cw_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("SEGM_CONV_WIDE=%s run %s: step ms %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done 2>&1 | tee gpurun_out/r05_conv_wide_step.log
