#!/bin/bash
# Round 4, closing call: the whole -m gpu suite on the closing commit, the bench line (default invocation), the step profile,
# the scan kernel summaries and the HBM traffic counters of the forward scan.
mkdir -p gpurun_out
echo "== gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | tail -25 | tee gpurun_out/r04_gpu_tests_final.log
echo "== bench"
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_final.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_final.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"]["ms"], r["backward"]["frac"], "| 3dir", {k: r["three_directions_per_launch"][k] for k in ("fwd_ms", "bwd_ms")})
print("config1", d.get("config1")); print("config4", d.get("config4")); print("cpu_baseline", d.get("cpu_baseline"))
PY
echo "== step profile"
bash tools/gpu_step_profile.sh r04_final 2>&1 | tail -50
echo "== scan kernels"
bash tools/gpu_scan_kernels.sh segmamba_amd/libsegmamba_hip.so r04final 2>&1 | tee gpurun_out/r04_scan_kernels_final.txt | tail -20
echo "== traffic"
COMMIT=$(cat .commit_for_traffic 2>/dev/null) bash tools/gpu_pmc_traffic.sh 2>&1 | tail -12
