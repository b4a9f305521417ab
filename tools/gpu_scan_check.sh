#!/bin/bash
# quick GPU check of the scan kernels: parity subset + per-kernel timings
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "regular_and_general or golden or half_precision or small_state or ragged" > gpurun_out/scan_check_tests.log 2>&1
tail -3 gpurun_out/scan_check_tests.log
bash tools/gpu_scan_prof.sh
