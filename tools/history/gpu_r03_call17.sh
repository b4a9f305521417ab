#!/bin/bash
# Round 3, GPU call 17 (after the ABI 4 additions): the driver's build() + smoke(), the scan / conv1d / multi-direction parity tests, a step.
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "amdgpu.ids\|MIOpen\|Gridwise" | tail -2
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_reference_bindings.py tests/test_gpu_dropin.py -m gpu -q -k "scan or conv1d or mamba or binding or inference or inner" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | cut -c1-200
