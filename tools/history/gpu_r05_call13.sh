#!/bin/bash
# Round 5, call 13: 16-byte stores in segm_pointwise_cf / segm_linear_rows: stand-alone, then the step with / without, interleaved
mkdir -p gpurun_out
(SEGM_POINTWISE_WIDE=0 python tools/history/r05/pointwise_wide.py; SEGM_POINTWISE_WIDE=1 python tools/history/r05/pointwise_wide.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_pointwise_wide.log
for i in 1 2 3; do
  for f in 1 0; do
    SEGM_POINTWISE_WIDE=$f SEGM_LINEAR_WIDE=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/wd_${f}_${i}.json
    python - $f $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/wd_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("SEGM_POINTWISE_WIDE = SEGM_LINEAR_WIDE = %s run %s: step ms %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done 2>&1 | tee gpurun_out/r05_wide_step.log
