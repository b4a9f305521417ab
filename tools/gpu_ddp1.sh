#!/bin/bash
# the data-parallel path on ONE GPU (world size 1 over RCCL): rendezvous, parameter broadcast, gradient all-reduce, graph step.
#   tools/gpu_ddp1.sh [flat|torch]
mkdir -p gpurun_out; MODE=${1:-flat}
SEGM_DDP=$MODE SEGM_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-configs > gpurun_out/ddp1_$MODE.log 2>&1
echo "$MODE rc=$?"; grep -v "amdgpu.ids\|MIOpen" gpurun_out/ddp1_$MODE.log | tail -2 | cut -c1-1500
