#!/bin/bash
# Round 3, GPU call 15: the driver's smoke() entry, and the data-parallel path (both modes) with world size 1.
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "amdgpu.ids\|MIOpen\|Gridwise" | tail -3
bash tools/gpu_ddp1.sh flat
bash tools/gpu_ddp1.sh torch
