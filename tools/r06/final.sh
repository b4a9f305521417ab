#!/bin/bash
# Round 6 closing set: the default bench line, the per-kernel step table, rocprofv3 --kernel-trace --stats of the roofline leg
mkdir -p gpurun_out
timeout 1200 python bench.py 2>gpurun_out/r06_bench_final7.err | tail -1 > gpurun_out/r06_bench_final7.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_final7.json"))
print("step ms", d["ms_per_step"], "value", d["value"], "launch", d["config"]["launch"])
print("mem", {k: v for k, v in d["config"]["peak_mem_mb"].items() if k.endswith("_mb")})
r = d["roofline"]; print("roofline frac", r["frac"], "ms", r["ms"], "bwd", r["backward"]["ms"], "3dir", r["three_directions_per_launch"]["fwd_ms"], r["three_directions_per_launch"]["bwd_ms"], "floor frac", r["valu"]["floor"])
print("fp32", d["roofline_fp32"]["frac"], d["roofline_fp32"]["ms"])
print("launch_forms", d["config"].get("launch_forms")); print("dropin", json.dumps(d.get("dropin_step"))[:600])
print("config1", json.dumps(d.get("config1"))[:300]); print("config4", json.dumps(d.get("config4"))[:500]); print("inference", d.get("inference"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:700])
PY
bash tools/gpu_step_profile.sh r06_final7 2>&1 | tail -32
bash tools/r06/scan_stats.sh 2>&1 | tail -30 | tee gpurun_out/r06_scan_stats3.txt
