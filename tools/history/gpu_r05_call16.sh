for i in 1 2; do for f in 0 1; do
  SEGM_WGRAD_GEMM_TN=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SEGM_WGRAD_GEMM_TN=$f run $i: step ms', d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_wgemm_tn_step.log
