#!/bin/bash
# HBM-side traffic of the scan kernels (rocprofv3 PMC; one counter per pass as the TCC block only has 4 slots) plus two
# calibration kernels with known byte counts: transpose_add (16-byte accesses) and conv1d_fwd (2-byte-per-lane rows).
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cat > /tmp/pmc.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, D, N, Lq = 2, 96, 16, 64 ** 3
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
delta = (0.5 * torch.rand(B, Lq, D, device=dev, generator=g)).bfloat16()
A = -0.5 * torch.rand(D, N, device=dev, generator=g)
Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
Dv = torch.randn(D, device=dev, generator=g); db = 0.5 * torch.rand(D, device=dev, generator=g)
for _ in range(3):
    f = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True)
    ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True, chunk=f["chunk"])
    ops_raw.transpose_add(hip, u)                                   # calibration: 2 * 100.7 MB
    w = torch.randn(D, 4, device=dev, generator=g)
    ops_raw.conv1d_fwd(hip, u, w, None, True, channel_last=True)    # calibration: 2 * 100.7 MB
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/prof/pmc_$c -o pmc -- python /tmp/pmc.py > gpurun_out/prof_pmc_$c.log 2>&1
  echo "$c rc=$?"; ls gpurun_out/prof/pmc_$c | head
done
