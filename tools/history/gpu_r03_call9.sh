#!/bin/bash
# Round 3, GPU call 9: the skinny TN kernel at the four stages' shapes, the default bench line (graph step, config1 / config4, roofline with
# three directions per launch, CPU baseline), and the per-kernel table of one step.
mkdir -p gpurun_out
echo "== skinny_tn parity"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "skinny or wgrad_gemm" 2>&1 | tail -3
echo "== A/B skinny route"
for v in 0 1; do echo "SEGM_SKINNY_TN=$v"; SEGM_SKINNY_TN=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline --no-graph 2>/dev/null | cut -c1-200; done
echo "== default bench"
timeout 900 python bench.py > gpurun_out/r03_call9_bench.json 2> gpurun_out/r03_call9_bench.err; echo "rc=$?"
cut -c1-400 gpurun_out/r03_call9_bench.json; tail -3 gpurun_out/r03_call9_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03_call9_bench.json").read().strip().splitlines()[-1])
    print(json.dumps(d["roofline"], indent=None)[:1500]); print(json.dumps(d["config"])[:1200]); print(d.get("cpu_baseline"))
    for k in ("config1_mamba_block", "config4_long_scan", "ddp"):
        print(k, json.dumps(d.get(k))[:600])
except Exception as e:
    print("parse failed", e)
PY
echo "== step profile"
bash tools/gpu_step_profile.sh r03_step_v4 2>&1 | tail -40
