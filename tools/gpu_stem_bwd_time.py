import os, sys, torch
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.rand(2, 4, 128, 128, 128, device="cuda"); w = 0.05 * torch.randn(48, 4, 7, 7, 7, device="cuda"); dy = torch.randn(2, 48, 64, 64, 64, device="cuda")
def wg(x, w, dy):
    return torch.ops.aten.convolution_backward(dy, x, w, [48], [2, 2, 2], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1, [False, True, True])
for name, cast in (("bf16 NCDHW", lambda t: t.bfloat16()), ("fp32 NCDHW", lambda t: t), ("bf16 NDHWC", lambda t: t.bfloat16().contiguous(memory_format=torch.channels_last_3d)),
                   ("fp16 NCDHW", lambda t: t.half())):
    a, b, c = cast(x), cast(w), cast(dy)
    try:
        print(f"stem weight gradient, {name}: {timeit(lambda: wg(a, b, c)) * 1e3:.0f} us", flush=True)
    except Exception as e:
        print(name, "failed:", str(e)[:100])
# as a GEMM on an explicit space-to-depth + unfold of the SMALL operand: dW = dy (48 x V) @ cols (V x 1372); cols built by unfold = 1.4 GB (what MIOpen does)
# alternative: correlate per tap with strided views (no column matrix): 343 einsum calls are too many; 7 calls over kz with a (ky, kx) unfold of one z-slab
xb, dyb = x.bfloat16(), dy.bfloat16()
xp = torch.nn.functional.pad(xb, (3, 3, 3, 3, 3, 3))
def wg_views():
    out = torch.empty(48, 4, 7, 7, 7, device="cuda", dtype=torch.float32)
    dyf = dyb.reshape(2, 48, -1)
    for kz in range(7):
        for ky in range(7):
            v = xp[:, :, kz:kz + 128:2, ky:ky + 128:2, :]                     # (2, 4, 64, 64, 134)
            cols = v.unfold(4, 7, 2)                                          # (2, 4, 64, 64, 64, 7) view
            out[:, :, kz, ky, :] = torch.einsum("bcv,bivk->cik", dyf.float(), cols.reshape(2, 4, -1, 7).float())
    return out
try:
    print(f"stem weight gradient, 49 einsums on strided views (fp32): {timeit(wg_views, reps=2, warm=1) * 1e3:.0f} us", flush=True)
except Exception as e:
    print("views failed:", str(e)[:200])
