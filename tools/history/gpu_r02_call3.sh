#!/bin/bash
# Round 2, third GPU call: new tests (fp16 loop, augmenter, predictor), the untimed convolution variants as autotune candidates,
# fp16 step time.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_predictor.py -m gpu -q -k "fp16 or augmenter or sliding_window or training_gradients" > gpurun_out/r02_gpu_tests3.log 2>&1; tail -6 gpurun_out/r02_gpu_tests3.log
for e in "SEGM_CONV_FWD_UNTIMED=0" "SEGM_CONV_FWD_UNTIMED=1" "SEGM_AMP=fp16"; do echo "== $e"; env $e SEGM_CONV_VERBOSE=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline --steps 10 > gpurun_out/r02_bench_$e.out 2> gpurun_out/r02_bench_$e.err; tail -1 gpurun_out/r02_bench_$e.out | cut -c1-200; grep -h "conv3d autotune" gpurun_out/r02_bench_$e.out gpurun_out/r02_bench_$e.err | cut -c1-260 | tail -12; done 2>&1 | tee gpurun_out/r02_bench_variants.log
