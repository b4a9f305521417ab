#!/bin/bash
# Round 3, GPU call 14: (1) the fp32 L = 2^24 reversed case with the forward kernels of commit 85d3d9c and with the current ones (did the
# ring / flag rewrite change the rounding?), (2) segm_channel_sum parity + the model tests, (3) step time with / without it.
mkdir -p gpurun_out
for v in $GRAFT_REPO_ROOT/build/variants/cur_oldfwd.so ""; do
  echo "== fp32 2^24 reversed, forward kernels: ${v:-current}"
  rm -f gpurun_out/parity_log.jsonl
  SEGM_LIB_OUT=$v timeout 600 python -m pytest tests/test_gpu_at_size.py -m gpu -q -k "fp32_reversed" 2>&1 | tail -2
  grep "fp32 reversed" gpurun_out/parity_log.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['what'], 'max_abs_err %.3e' % d['max_abs_err'], 'worst %.3f' % d['worst'])"
done 2>&1 | tee gpurun_out/r03_fp32_2p24_ab.log
echo "== channel_sum + model tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "channel_sum or model or network or segmamba or inner or bimamba" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -3
echo "== bench"
for v in 0 1; do echo "SEGM_CHANNEL_SUM_HIP=$v"; SEGM_CHANNEL_SUM_HIP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200; done
