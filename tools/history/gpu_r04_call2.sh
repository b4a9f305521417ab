#!/bin/bash
# Round 4, GPU call 2: the backward main kernel as independent waves (per-d-tile slabs of dB / dC + fixed-order sum kernel),
# two waves per SIMD with the rows of the next window prefetched; assembly DPP stages against the intrinsic form.
mkdir -p gpurun_out
echo "== scan parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_at_size.py -m gpu -q -x -k "scan" 2>&1 | tail -3 | tee gpurun_out/r04_call2_parity.log
echo "== variants"
timeout 600 python tools/gpu_scan_ab.py segmamba_amd/libsegmamba_hip.so build/variants/r04_noasm.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_scan_ab2.log
echo "== kernels"
bash tools/gpu_scan_kernels.sh segmamba_amd/libsegmamba_hip.so r04b 2>&1 | tee gpurun_out/r04_scan_kernels_b.txt
echo "== model parity + step"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r04_call2_model.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r04_bench_call2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_call2.json"))
r = d["roofline"]
print("step ms", d["ms_per_step"], "vol/s", d["value"], "| scan fwd ms", r["ms"], "frac", r["frac"], "| bwd", r["backward"], "| 3dir", {k: r["three_directions_per_launch"][k] for k in ("fwd_ms", "bwd_ms")})
PY
