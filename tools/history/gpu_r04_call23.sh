#!/bin/bash
# Round 4, GPU call 23: depth-to-space / space-to-depth kernel behind the k2s2 transposed convolutions: parity, timing vs ATen's permute, step A/B
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "depth_to_space" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -2
timeout 200 python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu
hip = L.get_lib()
for (B, C, D) in ((2, 48, 64), (2, 96, 32), (2, 192, 16)):
    blk = torch.randn(B, C * 8, D, D, D, device="cuda").bfloat16()
    t0 = time_gpu(lambda: blk.reshape(B, C, 2, 2, 2, D, D, D).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, C, 2 * D, 2 * D, 2 * D), 20)
    t1 = time_gpu(lambda: ops_raw.depth_to_space2(hip, blk), 20)
    vol = ops_raw.depth_to_space2(hip, blk)
    t2 = time_gpu(lambda: ops_raw.space_to_depth2(hip, vol), 20)
    gb = 2 * blk.numel() * 2 / 1e9
    print(f"depth-to-space {B}x{C}x{D}^3 -> {2 * D}^3: ATen permute {t0 * 1e3:6.1f} us ({gb / t0 * 1e3:.0f} GB/s)  kernel {t1 * 1e3:6.1f} us ({gb / t1 * 1e3:.0f} GB/s)  inverse {t2 * 1e3:6.1f} us", flush=True)
PY
echo "== step"
for v in 1 0 1 0; do echo "SEGM_D2S_HIP=$v"; SEGM_D2S_HIP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; done
} | tee gpurun_out/r04_d2s.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_network_bf16.py -m gpu -q -x -k "segmamba or network" 2>&1 | grep -v "GridwiseOp\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04_d2s.log
