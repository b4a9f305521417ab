#!/bin/bash
# GPU durations (rocprofv3 kernel trace) of linear_rows vs the vendor GEMM on the small projection shapes
mkdir -p gpurun_out/prof; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof/r06_linsmall -o t -- python $R/tools/gpu_linear_small_time.py > /dev/null 2>&1
python3 - $R/gpurun_out/prof/r06_linsmall <<'PY' | tee $R/gpurun_out/r06_linear_small_gpu.log
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of the same kernel name = one timing loop
runs = []
for r in rows:
    n = r["Kernel_Name"][:70]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == n: runs[-1][1].append(d)
    else: runs.append([n, [d]])
for n, v in runs:
    if len(v) >= 50: print("%7.1f us median (n=%3d)  %s" % (statistics.median(v), len(v), n))
PY
