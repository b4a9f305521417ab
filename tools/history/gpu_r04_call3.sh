#!/bin/bash
# Round 4, GPU call 3: checkpoints state-fastest (16-byte stores forward, 8-byte loads backward); nothing inside the backward
# state-pair loop comes from memory (A and the window checkpoint in LDS).
mkdir -p gpurun_out
echo "== scan parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -3 | tee gpurun_out/r04_call3_parity.log
echo "== variants"
timeout 600 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_scan_ab3.log
echo "== kernels"
bash tools/gpu_scan_kernels.sh segmamba_amd/libsegmamba_hip.so r04c 2>&1 | tee gpurun_out/r04_scan_kernels_c.txt
echo "== model parity + step"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r04_call3_model.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r04_bench_call3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_call3.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"], "| 3dir", {k: r["three_directions_per_launch"][k] for k in ("fwd_ms", "bwd_ms")})
PY
