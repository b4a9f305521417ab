#!/bin/bash
# Round 6 closing set on the closing tree: GPU suite, smoke, then tools/r06/final.sh (bench line, step table, scan stats)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -4 | tee gpurun_out/r06_gpu_tests_final7.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen(HIP)" | tail -3 | tee gpurun_out/r06_smoke_final7.log
bash tools/r06/final.sh
