#!/bin/bash
mkdir -p gpurun_out
SEGM_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/ddp1.log 2>&1
echo rc=$?; grep -v "amdgpu.ids\|MIOpen" gpurun_out/ddp1.log | tail -4 | cut -c1-400
