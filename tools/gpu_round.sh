#!/bin/bash
# One GPU session: quick kernel sanity + roofline, the GPU test-suite, smoke, full bench, kernel-trace profile.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sanity A"; SEGM_QUICK=1 SEGM_TAG=a timeout 300 python tools/gpu_sanity.py > gpurun_out/sanity_a.log 2>&1; echo "rc=$?"; grep -v MIOpen gpurun_out/sanity_a.log | tail -14
if [ -f build/libsegmamba_hip_w2.so ]; then echo "== sanity B (bwd capped at 2 waves/SIMD)"; SEGM_LIB=build/libsegmamba_hip_w2.so SEGM_QUICK=1 SEGM_TAG=b timeout 300 python tools/gpu_sanity.py > gpurun_out/sanity_b.log 2>&1; echo "rc=$?"; grep -v MIOpen gpurun_out/sanity_b.log | tail -12; fi
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v MIOpen gpurun_out/pytest_gpu.log | tail -15
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 4 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log | cut -c1-1500
