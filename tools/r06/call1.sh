#!/bin/bash
# Round 6, call 1: per-wave timeline of the forward scan passes (timeline build), then the default bench line (now with peak memory)
mkdir -p gpurun_out
timeout 600 python tools/gpu_scan_timeline.py build/variants/libsegm_timeline.so gpurun_out/r06_scan_wave_timeline.txt 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -150
timeout 900 python bench.py 2>gpurun_out/r06_bench_call1.err | tail -1 > gpurun_out/r06_bench_call1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_call1.json"))
print("step ms", d["ms_per_step"], "value", d["value"], "mem", d["config"].get("peak_mem_mb"), "inference", d.get("inference"))
print("roofline frac", d["roofline"]["frac"], "ms", d["roofline"]["ms"], "3dir", d["roofline"]["three_directions_per_launch"])
PY
