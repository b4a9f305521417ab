#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sanity"; SEGM_QUICK=1 SEGM_TAG=c timeout 300 python tools/gpu_sanity.py > gpurun_out/sanity_c.log 2>&1; echo "rc=$?"; grep -v MIOpen gpurun_out/sanity_c.log | grep -E "262144|32768" 
echo "== bench NCDHW"; timeout 900 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/bench_ncdhw.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_ncdhw.log | cut -c1-400
echo "== bench NDHWC"; SEGM_CHANNELS_LAST_3D=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/bench_ndhwc.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_ndhwc.log | cut -c1-400
echo "== pytest kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "not reference_matrix and not repeatability" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v MIOpen gpurun_out/pytest_gpu.log | tail -5
