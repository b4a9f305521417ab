#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -X faulthandler - <<'PY' 2>&1 | grep -v "MIOpen(HIP)\|GridwiseOp\|amdgpu.ids" | tail -40
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(json.dumps(bench.dropin_step(dev, 128, 2, 10, 3), indent=1), flush=True)
PY
echo "rc=$?"
