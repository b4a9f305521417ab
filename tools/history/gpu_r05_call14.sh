(SEGM_POINTWISE_NT=0 python tools/history/r05/pointwise_wide.py; SEGM_POINTWISE_NT=1 python tools/history/r05/pointwise_wide.py; SEGM_LINEAR_NT=0 python tools/history/r05/linear_wide.py; SEGM_LINEAR_NT=1 python tools/history/r05/linear_wide.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_wide_nt.log
for i in 1 2; do for f in 1 0; do
  SEGM_POINTWISE_NT=$f SEGM_LINEAR_NT=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NT=$f run $i: step ms', d['ms_per_step'])"
done; done 2>&1 | tee -a gpurun_out/r05_wide_nt.log
