#!/bin/bash
# Round 3, final GPU call: the whole -m gpu suite, the default bench line, the per-kernel table of one step, the scan kernels'
# durations at the roofline shape (rocprofv3 --kernel-trace --stats), the HBM-side traffic of one forward scan launch (PMC passes).
mkdir -p gpurun_out; rm -f gpurun_out/parity_log.jsonl
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "GridwiseOp\|MIOpen(HIP)" > gpurun_out/r03_gpu_tests_final.log; tail -16 gpurun_out/r03_gpu_tests_final.log
timeout 900 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"; tail -1 gpurun_out/r03_bench_final.json | cut -c1-600
bash tools/gpu_step_profile.sh r03_step_final 2>&1 | tail -12
cp segmamba_amd/libsegmamba_hip.so build/variants/r03_final.so
bash tools/gpu_scan_kernels.sh build/variants/r03_final.so r03_final 2>&1 | tee gpurun_out/r03_scan_kernels_final.txt | tail -16
cp gpurun_out/prof/scan_r03_final/*kernel_stats.csv gpurun_out/r03_scan_kernel_stats_final.csv 2>/dev/null
COMMIT=$(cat .commit_for_traffic 2>/dev/null || echo unknown) bash tools/gpu_pmc_traffic.sh 2>&1 | tail -12
