#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/pytest_gpu_full.log 2>&1
echo "rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/pytest_gpu_full.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
