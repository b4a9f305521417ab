#!/bin/bash
# Round 6, call 7: the N > 1 code path on one rank (SEGM_FORCE_DDP=1 through the driver's launcher): the other launch form behind the
# timed region, and its watchdog (a 1 ms limit: rank 0 must still print the line and exit 0)
mkdir -p gpurun_out
SEGM_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-configs > gpurun_out/r06_ddp1_flat.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|MIOpen" gpurun_out/r06_ddp1_flat.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'ddp', json.dumps(d['config']['ddp'])[:900])"
SEGM_BENCH_OTHER_FORM_TIMEOUT_S=0.001 SEGM_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-configs > gpurun_out/r06_ddp1_watchdog.log 2>&1
echo "watchdog rc=$?"; grep -v "amdgpu.ids\|MIOpen" gpurun_out/r06_ddp1_watchdog.log | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'other_form', d['config']['ddp']['other_form'])"
