#!/bin/bash
# Round 4, GPU call 4: the whole -m gpu suite (with the conditioned per-block tests, the at-size convolution backward, fp16 network),
# scan kernels with pair-layout checkpoints, step profile, bench line, HBM traffic counters of the forward scan.
mkdir -p gpurun_out
echo "== gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/r04_gpu_tests_call4.log
echo "== scan"
timeout 600 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_scan_ab4.log
bash tools/gpu_scan_kernels.sh segmamba_amd/libsegmamba_hip.so r04d 2>&1 | tee gpurun_out/r04_scan_kernels_d.txt
echo "== step profile"
bash tools/gpu_step_profile.sh r04_v1 2>&1 | tail -45
echo "== bench"
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r04_bench_call4.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_call4.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "floor frac", r["valu"]["floor"], "| bwd", r["backward"]["ms"], r["backward"]["frac"], "| 3dir", {k: r["three_directions_per_launch"][k] for k in ("fwd_ms", "bwd_ms")})
print("config1", d.get("config1")); print("config4", d.get("config4"))
PY
echo "== traffic"
COMMIT=$(cat .commit_for_traffic 2>/dev/null) bash tools/gpu_pmc_traffic.sh 2>&1 | tail -12
