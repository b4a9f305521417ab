#!/bin/bash
# One GPU session: quick kernel sanity + roofline, the GPU test-suite, smoke, full bench, kernel-trace profile.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sanity" ; timeout 300 python tools/gpu_sanity.py > gpurun_out/sanity.log 2>&1; echo "sanity rc=$?"; tail -15 gpurun_out/sanity.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 4 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log | cut -c1-1500
