#!/bin/bash
# GPU session: rocprofv3 kernel-trace summaries (per-kernel time) for the bench step and the scan micro-benchmark.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== rocprof bench"; timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/prof_bench.log | cut -c1-600
echo "== rocprof scan"; SEGM_QUICK=1 SEGM_TAG=p timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/scan -o scan -- python tools/gpu_sanity.py > gpurun_out/prof_scan.log 2>&1; echo "rc=$?"
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "--- $f"; head -40 $f | cut -c1-200; done
# keep only the small summaries (traces can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== pytest gpu (rest)"; timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v MIOpen gpurun_out/pytest_gpu.log | tail -12
