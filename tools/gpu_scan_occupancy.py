"""How the scan kernels' time grows with the number of waves a SIMD has to run: B=2, D=32 (one 32-channel tile, two chunks per wave -
the stage-0 kernel instantiation), chunk 256, L = k * 262144  ->  exactly k waves per SIMD (1024 SIMDs), every wave with the same
256 steps of work.  A kernel bound by VALU throughput takes k times as long with k waves; one bound by a lone wave's issue latency
stays flat until the pipe is full.  Prints the forward / backward call times; run under rocprofv3 --kernel-trace for the per-kernel
durations (tools/gpu_scan_occupancy.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu

if os.environ.get("SEGM_OCC_LIB"):                      # a variant build (tools/build_scan_ablations.py)
    L.LIB_PATH = os.path.abspath(os.environ["SEGM_OCC_LIB"])
hip = L.get_lib()
dev = torch.device("cuda")
dtype = torch.bfloat16
B, D, N = 2, 32, 16
for k in [int(x) for x in os.environ.get("SEGM_OCC_K", "1,2,3,4,5,6,8").split(",")]:
    Lq = 262144 * k
    g = torch.Generator(device=dev).manual_seed(k)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(dtype)
    u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
    delta = (0.5 * torch.rand(B, Lq, D, device=dev, generator=g)).to(dtype)
    A = -0.5 * torch.rand(D, N, device=dev, generator=g)
    Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
    Dv = torch.randn(D, device=dev, generator=g)
    db = 0.5 * torch.rand(D, device=dev, generator=g)
    fwd = lambda: ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True, chunk=256)
    f = fwd()
    bwd = lambda: ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True, chunk=256)
    ms_f, ms_b = time_gpu(fwd, 8), time_gpu(bwd, 4)
    print(f"waves/SIMD {k}: L {Lq:8d}  fwd {ms_f:7.4f} ms ({ms_f / k:6.4f} per wave round)   bwd {ms_b:7.4f} ms ({ms_b / k:6.4f})", flush=True)
    del u, z, dout, delta, Bm, Cm, f
