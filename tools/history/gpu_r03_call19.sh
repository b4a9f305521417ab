#!/bin/bash
# Round 3, GPU call 19: does the timing tuner still agree with the frozen routing table?  (step time with each, and the tuner's timings per shape)
mkdir -p gpurun_out
for v in 0 1; do echo "SEGM_CONV_AUTOTUNE=$v"; SEGM_CONV_AUTOTUNE=$v SEGM_CONV_VERBOSE=$v timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-configs --no-roofline --no-graph 2> gpurun_out/r03_tuner_$v.err | cut -c1-200; done
grep -i "conv\|pick\|->" gpurun_out/r03_tuner_1.err | grep -v "MIOpen\|Gridwise" | head -80
