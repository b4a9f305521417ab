#!/bin/bash
# Round 4, the very last call: default bench line + per-kernel step table on the final commit (after the depth-to-space kernels)
mkdir -p gpurun_out
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_final3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_final3.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"])
PY
bash tools/gpu_step_profile.sh r04_final3 2>&1 | tail -40
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "depth_to_space or dt_proj or conv3" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -2
