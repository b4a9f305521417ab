#!/bin/bash
# Round 3, GPU call 18: SQ issue / stall counters of the final scan kernels, and the default bench line on the closing commit.
mkdir -p gpurun_out build/variants
cp segmamba_amd/libsegmamba_hip.so build/variants/r03_closing.so
bash tools/gpu_scan_pmc.sh build/variants/r03_closing.so r03_closing 2>&1 | tee gpurun_out/r03_scan_pmc_final.txt | tail -30
timeout 900 python bench.py > gpurun_out/r03_bench_closing.json 2> gpurun_out/r03_bench_closing.err; echo "bench rc=$?"; tail -1 gpurun_out/r03_bench_closing.json | cut -c1-300
