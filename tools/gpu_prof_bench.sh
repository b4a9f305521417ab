#!/bin/bash
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/bench15 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench15.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/prof_bench15.log | cut -c1-300
ls -la gpurun_out/prof/bench15
