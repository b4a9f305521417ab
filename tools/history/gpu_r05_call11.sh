#!/bin/bash
# Round 5, call 11: conv1 + conv3 of a residual block as one node (conv3d._ResFront): the block / network parity tests, then the step
# with and without it (SEGM_RES_FRONT), interleaved
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -15 | tee gpurun_out/r05_res_front_tests.log
for i in 1 2 3; do
  for f in 1 0; do
    SEGM_RES_FRONT=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/rf_${f}_${i}.json
    python - $f $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/rf_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("SEGM_RES_FRONT=%s run %s: step ms %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done 2>&1 | tee gpurun_out/r05_res_front_step.log
