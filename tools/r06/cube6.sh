#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cube" 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_blocks_conditioned.py -q -m gpu 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" | tail -2
for d in 1 0 1 0; do SEGM_CUBE_DIRECT=$d timeout 600 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct $d', d['ms_per_step'], d['inference']['ms_per_case'])"; done
