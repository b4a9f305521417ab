#!/bin/bash
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cat > /tmp/sp.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from bench import scan_roofline
r = scan_roofline(torch.bfloat16, torch.device("cuda"))
print(r["ms"], r["backward"]["ms"])
r = scan_roofline(torch.float32, torch.device("cuda"))
print(r["ms"], r["backward"]["ms"])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/scan11 -o scan -- python /tmp/sp.py > gpurun_out/prof_scan11.log 2>&1
tail -3 gpurun_out/prof_scan11.log
