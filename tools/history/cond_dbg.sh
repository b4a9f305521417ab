#!/bin/bash
run() { echo "== $*"; for i in 1 2 3; do env "$@" timeout 600 python -m pytest tests/test_gpu_blocks_conditioned.py -m gpu -q -k "unet_res_block_conditioned or gsc or up_block" 2>&1 | grep "AssertionError\|passed\|failed" | cut -c1-400; done; }
run A=1
run SEGM_WGRAD_V1=1
run SEGM_CONV_CHAIN_VAR=0
run SEGM_CONV_STATS=0
run SEGM_CONV_STATS=0 SEGM_WGRAD_V1=1 SEGM_CONV_CHAIN_VAR=0
