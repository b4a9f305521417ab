#!/bin/bash
# Round 5, closing call: build check, smoke, the whole GPU suite (with durations), the driver's bench command
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -2 | tee gpurun_out/r05_smoke_final3.log
timeout 1800 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -22 | tee gpurun_out/r05_gpu_tests_final3.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r05_bench_final3.err | tail -1 > gpurun_out/r05_bench_final3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_final3.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"])
print("launch_forms", {k: v for k, v in d["config"].get("launch_forms", {}).items() if k.endswith("_ms")})
print("dropin", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("dropin_step", {}).items()})
PY
grep "bench.py +" gpurun_out/r05_bench_final3.err
bash tools/gpu_step_profile.sh r05_final3 2>&1 | tail -32
