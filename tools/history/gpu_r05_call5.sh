#!/bin/bash
# Round 5, call 5: shipped wgrad (ct) on all step shapes, conv / train-loop parity tests, the driver's bench command with dropin_step, step profile
mkdir -p gpurun_out
timeout 600 python tools/history/r05/wgrad_ab.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tee gpurun_out/r05_wgrad_ab2.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks_conditioned.py tests/test_gpu_model.py -m gpu -q -x -k "wgrad or conv3 or fp16 or sgd or clip or train" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -5 | tee gpurun_out/r05_call5_tests.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r05_bench_call5.err | tail -1 > gpurun_out/r05_bench_call5.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call5.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "ceiling", r.get("ceiling"), "| bwd", r["backward"]["ms"], r["backward"]["frac"], r["backward"].get("floor"))
print("gpu_state", d["config"].get("gpu_state"))
print("dropin", json.dumps(d.get("dropin_step"), indent=1))
PY
tail -3 gpurun_out/r05_bench_call5.err
bash tools/gpu_step_profile.sh r05_call5 2>&1 | tail -45
