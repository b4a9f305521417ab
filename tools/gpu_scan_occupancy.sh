#!/bin/bash
# per-kernel durations of tools/gpu_scan_occupancy.py by grid size:  tools/gpu_scan_occupancy.sh <tag>   (env passes through)
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof/occ_$1 -o occ -- python $R/tools/gpu_scan_occupancy.py 2>&1 | grep -E "waves/SIMD|Error|error"
python3 - $R/gpurun_out/prof/occ_$1 <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "segm" in n and ("agg" in n or "apply" in n or "main" in n):
        short = n.split("segm")[1][:34]
        waves = int(r["Grid_Size_X"]) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) // 64
        d[(short, waves)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, w), v in sorted(d.items()):
    m = statistics.median(v)
    print(f"{k:36s} waves/SIMD {w / 1024:4.1f}  {m:8.1f} us   {m / (w / 1024):7.1f} us per wave round   {m * 2.1e3 / (w / 1024) / 256:6.0f} cycles per wave-step at 2.1 GHz")
PY
