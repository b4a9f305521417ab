#!/bin/bash
# end-of-round validation: full GPU test suite, smoke, bench (default arguments), step profile
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu_full.log 2>&1
echo "pytest rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/pytest_gpu_full.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?"
grep -v "amdgpu.ids\|MIOpen" gpurun_out/final_bench.log | tail -1 | cut -c1-1200
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/final -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_final.log 2>&1; echo "prof rc=$?"
bash tools/gpu_scan_prof.sh > /dev/null 2>&1
