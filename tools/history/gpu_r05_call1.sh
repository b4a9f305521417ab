#!/bin/bash
# Round 5, call 1: wgrad A/B (round-5 row loop vs v1), its parity tests, the driver's bench command, a step profile
mkdir -p gpurun_out
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 > gpurun_out/r05_clocks_before.txt
timeout 600 python tools/history/r05/wgrad_ab.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tee gpurun_out/r05_wgrad_ab.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py -m gpu -q -x -k "wgrad or conv3" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee gpurun_out/r05_call1_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05_bench_call1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call1.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"])
PY
SEGM_WGRAD_V1=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_call1_v1.json
python -c "import json; d=json.load(open('gpurun_out/r05_bench_call1_v1.json')); print('v1 wgrad step ms', d['ms_per_step'])"
bash tools/gpu_step_profile.sh r05_call1 2>&1 | tail -45
