#!/bin/bash
# Round 4, last call: build() + smoke(), the whole -m gpu suite and the default bench line on the final commit
mkdir -p gpurun_out
echo "== build + smoke"
timeout 600 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | tail -3 | tee gpurun_out/r04_smoke_final2.log
echo "== gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | tail -12 | tee gpurun_out/r04_gpu_tests_final2.log
echo "== bench"
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_final2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_final2.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"], "| traffic at", r["traffic_source"]["measured_at"])
PY
