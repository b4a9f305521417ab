#!/bin/bash
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== bench (autotuned convs)"; SEGM_CONV_VERBOSE=1 timeout 900 python bench.py --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; echo "rc=$?"; grep "autotune" gpurun_out/bench4.log | cut -c1-200; tail -1 gpurun_out/bench4.log | cut -c1-700
echo "== profile"; timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/bench2 -o bench -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench2.log 2>&1; echo "rc=$?"
echo "== model tests"; timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/pytest_model.log 2>&1; echo "rc=$?"; grep -v MIOpen gpurun_out/pytest_model.log | tail -5
