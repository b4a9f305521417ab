#!/bin/bash
# per-kernel durations of the scan launches at the roofline shape for one library variant: tools/gpu_scan_kernels.sh <lib.so> <tag>
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/scan_$2 -o scan -- python $R/tools/gpu_scan_ab.py $R/$1 > $R/gpurun_out/prof/scan_$2.log 2>&1
python3 - $R/gpurun_out/prof/scan_$2 <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "segm" in k:
        print(f"{statistics.median(v):9.1f} us median  n={len(v):4d}  {k}")
PY
