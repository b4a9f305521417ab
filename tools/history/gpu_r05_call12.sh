#!/bin/bash
# Round 5, call 12: InstanceNorm passes streamed by tensor size (non-temporal, four packets in flight; csrc/instnorm.hip norm_mode):
# the norm / block / network parity tests, then the step with the size rule (default) and with mode 0 forced, interleaved
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "norm or block or segmamba or graphed or res_front or up_block or gsc" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -6 | tee gpurun_out/r05_inorm_stream_tests.log
for i in 1 2 3; do
  for f in auto 0; do
    if [ $f = auto ]; then unset SEGM_NORM_NT; else export SEGM_NORM_NT=$f; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/ns_${f}_${i}.json
    python - $f $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/ns_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("SEGM_NORM_NT=%s run %s: step ms %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done 2>&1 | tee gpurun_out/r05_inorm_stream_step.log
