import sys, torch
sys.path.insert(0, ".")
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for B, Cin, Cout, S in [(2, 48, 48, 128 ** 3), (2, 4, 48, 128 ** 3), (2, 48, 4, 128 ** 3), (2, 48, 96, 64 ** 3), (2, 96, 48, 64 ** 3), (2, 96, 96, 32 ** 3)]:
    x = torch.randn(B, Cin, S, device="cuda").bfloat16(); w = (0.1 * torch.randn(Cout, Cin, device="cuda")).bfloat16()
    b = torch.randn(Cout, device="cuda")
    def blas():
        y = torch.bmm(w.unsqueeze(0).expand(B, -1, -1), x); y += b.bfloat16().view(1, -1, 1); return y
    t_blas = timeit(blas); t_hip = timeit(lambda: ops_raw.pointwise_cf(hip, x, w, b))
    gb = B * S * (Cin + Cout) * 2 / 1e9
    print(f"pointwise {Cin}->{Cout} S={S}: bmm + bias {t_blas*1e3:.0f} us ({gb/t_blas*1e3:.0f} GB/s)  segm_pointwise_cf {t_hip*1e3:.0f} us ({gb/t_hip*1e3:.0f} GB/s)", flush=True)
# the 7^3 stride-2 stem convolution: MIOpen (F.conv3d) against segm_stem_conv_fwd
x = torch.rand(2, 4, 128, 128, 128, device="cuda").bfloat16(); w = (0.05 * torch.randn(48, 4, 7, 7, 7, device="cuda")).bfloat16()
b = torch.randn(48, device="cuda")
t_lib = timeit(lambda: torch.nn.functional.conv3d(x, w, b.bfloat16(), stride=2, padding=3), reps=10)
t_hip = timeit(lambda: ops_raw.stem_conv_fwd(hip, x, w, b), reps=10)
print(f"stem 7^3 s2 4->48 on 2x4x128^3: MIOpen {t_lib*1e3:.0f} us  segm_stem_conv_fwd {t_hip*1e3:.0f} us ({69.0/t_hip:.0f} TFLOP/s)", flush=True)
