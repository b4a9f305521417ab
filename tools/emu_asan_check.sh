#!/bin/bash
# TEST INFRASTRUCTURE: the kernel sources on the CPU emulation of HIP, built with AddressSanitizer + UBSan - `static __shared__`
# arrays become instrumented globals, so out-of-bounds LDS indexing (and wild global pointers into ASan-owned memory) is
# reported; UBSan adds misaligned vector accesses (a 16-byte LDS or global access off its alignment faults on the GPU) and
# integer overflow in the index arithmetic.  Runs small problems through every kernel family.   bash tools/emu_asan_check.sh
set -e
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
OUT=build/emu_asan
mkdir -p $OUT
FLAGS="-O1 -g -fsanitize=address,undefined,alignment -fno-sanitize-recover=undefined,alignment -fno-omit-frame-pointer -std=c++17 -fPIC -pthread -Itests/emu -Wno-unused-value -DSEGM_EMU=1"
OBJS=""
for f in segmamba_amd/csrc/*.hip tests/emu/hip_emu_runtime.cpp; do
  o=$OUT/$(basename $f).o
  $CLANG $FLAGS '-DSEGM_PIN_F32(x)=' '-DSEGM_SCHED_FENCE()=' '-DSEGM_PIN_F2(x)=' '-DSEGM_WAVE_LDS_SYNC()=hipemu::sync_wave()' -x c++ -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
$CLANG -shared -pthread -fsanitize=address,undefined -shared-libasan $OBJS -o $OUT/libsegmamba_emu_asan.so
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 python - <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from segmamba_amd import lib as L, ops_raw
emu = L.SegmLib("build/emu_asan/libsegmamba_emu_asan.so")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 96, 2, 3, 40, generator=g).bfloat16(); w = (0.1 * torch.randn(48, 96, 3, 3, 3, generator=g)).bfloat16()
ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
for kw in (dict(), dict(chain=True), dict(chain=True, pitch48=True), dict(chain32=True)):
    y = ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:, :48]), None, **kw)
    ops_raw.conv3d_k3_fwd(emu, x[:, 48:], ops_raw.pack_conv3d_weight(w[:, 48:]), None, out=y, accumulate=True, **kw)
    assert (y.float() - ref).abs().max() <= 3e-2 * float(ref.abs().max()), kw
    print("conv fwd", kw, "ok", flush=True)
xn = torch.randn(1, 4, 2, 3, 16, generator=g).bfloat16(); wn = (0.1 * torch.randn(32, 4, 3, 3, 3, generator=g)).bfloat16()
ops_raw.conv3d_k3_fwd(emu, xn, ops_raw.pack_conv3d_weight(wn)); print("conv fwd 12-wave kernel ok", flush=True)
dy = torch.randn(1, 48, 2, 3, 40, generator=g).bfloat16()
ops_raw.conv3d_k3_wgrad(emu, x[:, :48].contiguous(), dy, torch.float32); print("wgrad ok", flush=True)
a = torch.randn(70, 96, generator=g).bfloat16(); wl = torch.randn(52, 96, generator=g).bfloat16()
ops_raw.linear_rows(emu, a, wl, torch.randn(52)); print("linear_rows ok", flush=True)
ps = [torch.randn(n) for n in (5, 16385, 0, 100)]; gs = [torch.randn(n) for n in (5, 16385, 0, 100)]; ms = [torch.zeros(n) for n in (5, 16385, 0, 100)]
ops_raw.sgd_clip_step(emu, ps, gs, ms, 0.01, 0.9, 1e-4, True, 1.0); print("sgd ok", flush=True)
lg = torch.randn(2, 4, 50, generator=g).bfloat16(); lb = torch.randint(0, 4, (2, 50), generator=g)
ops_raw.cross_entropy(emu, lg, lb); print("cross entropy ok", flush=True)
cs = torch.randn(2, 8, 4); ops_raw.conv1d_update(emu, torch.randn(2, 8), cs, torch.randn(8, 4), None, True)
st = torch.randn(2, 8, 16); ops_raw.state_update(emu, st, torch.randn(2, 8), torch.randn(2, 8), -torch.rand(8, 16), torch.randn(2, 16), torch.randn(2, 16))
print("decode ok", flush=True)
xi = torch.randn(2, 3, 5, 6, 7, generator=g).bfloat16()
y, m, r = ops_raw.instnorm_fwd(emu, xi, None, "relu"); ops_raw.instnorm_bwd(emu, xi, xi, m, r, None, "relu"); print("instnorm ok", flush=True)
xl = torch.randn(2, 48, 40, generator=g).bfloat16()
yl, m2, r2 = ops_raw.layernorm_tokens_fwd(emu, xl, torch.ones(48), torch.zeros(48)); ops_raw.layernorm_tokens_bwd(emu, xl, yl, m2, r2, torch.ones(48)); print("layernorm ok", flush=True)
ops_raw.transpose_add(emu, torch.randn(2, 70, 33, generator=g).bfloat16()); print("transpose ok", flush=True)
from tests import helpers as H
c = H.scan_case(2, 96, 16, 64, dtype=torch.bfloat16, seed=1)
H.run_scan(emu, c, "cpu", True, L.TIME_INTERLEAVED, 8); print("scan (regular-shape kernels) ok", flush=True)
c = H.scan_case(1, 40, 12, 50, dtype=torch.float32, seed=2)
H.run_scan(emu, c, "cpu", False, L.TIME_REVERSED, 1); print("scan (general kernels) ok", flush=True)
c = H.scan_case(1, 16, 16, 16 * 70, dtype=torch.float32, seed=3)
H.run_scan(emu, c, "cpu", True, L.TIME_REVERSED, 1, chunk=16); print("scan with several carry segments ok", flush=True)
xp = torch.randn(2, 56, 128, generator=g).bfloat16()[:, 8:48]; wpw = (0.1 * torch.randn(52, 40, generator=g)).bfloat16()
ops_raw.pointwise_cf(emu, xp, wpw, torch.randn(52)); print("pointwise_cf ok", flush=True)
xs = torch.randn(1, 3, 2, 8, 64, generator=g).bfloat16(); ws = (0.1 * torch.randn(20, 3, 7, 7, 7, generator=g)).bfloat16()
ops_raw.stem_conv_fwd(emu, xs, ws, torch.randn(20)); print("stem_conv_fwd ok", flush=True)
dys = torch.randn(1, 20, 1, 4, 32, generator=g).bfloat16()
ops_raw.stem_conv_wgrad(emu, ops_raw.stem_channel_last4(xs), dys, 3); print("stem_conv_wgrad (7^3) ok", flush=True)
x3 = torch.randn(1, 4, 2, 8, 32, generator=g).bfloat16(); w3 = (0.1 * torch.randn(48, 4, 3, 3, 3, generator=g)).bfloat16()
ops_raw.stem_conv_fwd(emu, x3, w3, None)
dy3 = torch.zeros(1, 48, 2 * 8 * 32 + 64, dtype=torch.bfloat16)[:, :, :2 * 8 * 32].view(1, 48, 2, 8, 32)         # padded channel stride
ops_raw.stem_conv_wgrad(emu, ops_raw.stem_channel_last4(x3), dy3, 4, 3); print("thin-input 3^3 forward / wgrad ok", flush=True)
ta = torch.randn(1000, 35, generator=g).bfloat16(); tb = torch.randn(1000, 104, generator=g).bfloat16()[:, :100]
ops_raw.wgrad_gemm(emu, ta, tb, ops_raw.WGEMM_TN)
ta = torch.randn(300, 72, generator=g).bfloat16(); tb = torch.randn(300, 200, generator=g).bfloat16()
ops_raw.wgrad_gemm(emu, ta, tb, ops_raw.WGEMM_TN); print("wgrad_gemm TN ok", flush=True)
na = torch.randn(2, 48, 96, generator=g).bfloat16()[:, 8:]; nb = torch.randn(2, 20, 96, generator=g).bfloat16()
ops_raw.wgrad_gemm(emu, na, nb, ops_raw.WGEMM_NT); print("wgrad_gemm NT ok", flush=True)
xq = torch.full((2, 3, 5 * 6 * 8 + 24), float("nan"), dtype=torch.bfloat16)[:, :, :240].view(2, 3, 5, 6, 8)       # padded instances
xq.copy_(torch.randn(2, 3, 5, 6, 8, generator=g))
yq, mq, rq = ops_raw.instnorm_fwd(emu, xq, None, "leaky_relu"); ops_raw.instnorm_bwd(emu, xq, xq, mq, rq, None, "leaky_relu")
print("instnorm on padded instances ok", flush=True)
# round 6
xc = torch.zeros(1, 2, 2, 40, 56, dtype=torch.bfloat16)[:, :, :, :32, :48]                                       # padded voxel / row strides
xc.copy_(torch.randn(1, 2, 2, 32, 48, generator=g)); wc = (0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
for w8 in (False, True):
    yc = ops_raw.conv3d_k3_fwd_cl(emu, xc, ops_raw.conv3d_cl_weight_image(emu, wc), torch.randn(48), waves8=w8)
    ops_raw.conv3d_k3_fwd_cl(emu, xc, ops_raw.conv3d_cl_weight_image(emu, wc), None, out=yc, accumulate=True, waves8=w8)
print("channel-last 3^3 forward ok", flush=True)
a3 = [torch.randn(8 * 515, generator=g).bfloat16() for _ in range(3)]
ops_raw.add3(emu, *a3, out=a3[0]); print("add3 ok", flush=True)
ak = torch.randn(70, 400, generator=g).bfloat16()[:, 8:392]; wk = torch.randn(44, 384, generator=g).bfloat16()
yk = torch.zeros(70, 48, dtype=torch.bfloat16)[:, :44]
ops_raw.linear_rows(emu, ak, wk, torch.randn(44), out=yk); ops_raw.linear_rows(emu, ak, wk, None, out=yk, accumulate=True)
print("linear_rows K > 192 ok", flush=True)
ta = torch.randn(999, 40, generator=g).bfloat16()[:, :36]; tb = torch.randn(999, 48, generator=g).bfloat16()
ops_raw.wgrad_gemm(emu, ta, tb, ops_raw.WGEMM_TN); print("wgrad_gemm TN on padded rows / 48-column blocks ok", flush=True)
# round 6, second half: the cube kernels and the bank's gather
xq = torch.zeros(1, 65, 8, 16, 24, dtype=torch.bfloat16)[:, :64, :, :, :16]                                     # padded strides, two cubes in x and y
xq.copy_(torch.randn(1, 64, 8, 16, 16, generator=g)); wq = (0.1 * torch.randn(192, 64, 3, 3, 3, generator=g)).bfloat16()
for nt, sp in ((2, 1), (3, 2), (4, 0)):
    if 192 % (32 * nt) == 0:
        yq, stq = ops_raw.conv3d_k3_cube_fwd(emu, xq, ops_raw.conv3d_cube_weight_image(emu, wq), 192, torch.randn(192), nt=nt, splits=sp, want_stats=True)
        ops_raw.conv3d_k3_cube_fwd(emu, xq, ops_raw.conv3d_cube_weight_image(emu, wq), 192, None, out=yq, accumulate=True, nt=nt, splits=sp)
dyq = torch.randn(1, 192, 8, 16, 16, generator=g).bfloat16()
ops_raw.conv3d_k3_cube_fwd(emu, dyq, ops_raw.conv3d_cube_weight_image(emu, wq, flipped=True), 64)
print("cube forward / data gradient ok", flush=True)
ops_raw.conv3d_k3_cube_wgrad(emu, xq, dyq, torch.float32)                                                        # rows with x neighbours
ops_raw.conv3d_k3_cube_wgrad(emu, xq[:, :32, :, :8, :8], dyq[:, :64, :, :8, :8], torch.bfloat16)                # 8-wide: padding only
print("cube weight gradient ok", flush=True)
mq = ops_raw.conv3d_cube_index(emu, 192, 64, False, "cpu")
ops_raw.gather16(emu, wq.reshape(-1), mq, torch.empty(mq.numel(), dtype=torch.bfloat16))
ops_raw.gather16(emu, wq.reshape(-1), ops_raw.gather16_compact_map(mq), torch.empty(mq.numel(), dtype=torch.bfloat16), compact=True)
print("gather16 ok", flush=True)
print("AddressSanitizer run finished without reports")
PY
