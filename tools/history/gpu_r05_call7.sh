#!/bin/bash
# Round 5, call 7: the whole GPU suite + the driver's bench command (launch_forms, dropin_step)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -8 | tee gpurun_out/r05_gpu_tests_call7.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r05_bench_call7.err | tail -1 > gpurun_out/r05_bench_call7.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_call7.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"])
print("launch_forms", d["config"].get("launch_forms"))
print("dropin", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("dropin_step", {}).items()})
PY
