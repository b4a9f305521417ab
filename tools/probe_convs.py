"""GPU probe (not product): per-configuration time of SegMamba's 3-D convolutions through MIOpen (bf16, NCDHW),
forward / grad-input / grad-weight separately."""
import os, torch, torch.nn.functional as F
for k in ("FWD", "BWD", "WRW"): os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
dev = "cuda"
def tm(fn, it=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
cfgs = [  # (name, cin, cout, k, stride, pad, size)
    ("enc1.conv1 4->48 k3 @128", 4, 48, 3, 1, 1, 128), ("48->48 k3 @128 (x4)", 48, 48, 3, 1, 1, 128), ("dec2.conv1 96->48 k3 @128", 96, 48, 3, 1, 1, 128),
    ("stem 4->48 k7 s2 @128", 4, 48, 7, 2, 3, 128),
    ("gsc 48->48 k3 @64 (x2, bias)", 48, 48, 3, 1, 1, 64), ("enc2.conv1 48->96 k3 @64", 48, 96, 3, 1, 1, 64), ("96->96 k3 @64 (x2)", 96, 96, 3, 1, 1, 64), ("dec3.conv1 192->96 k3 @64", 192, 96, 3, 1, 1, 64),
    ("96->96 k3 @32 gsc(x2)", 96, 96, 3, 1, 1, 32), ("96->192 k3 @32", 96, 192, 3, 1, 1, 32), ("192->192 k3 @32 (x2)", 192, 192, 3, 1, 1, 32), ("384->192 k3 @32", 384, 192, 3, 1, 1, 32),
    ("192->192 k3 @16 gsc (x2)", 192, 192, 3, 1, 1, 16), ("192->384 k3 @16", 192, 384, 3, 1, 1, 16), ("384->384 k3 @16 (x2)", 384, 384, 3, 1, 1, 16), ("768->384 k3 @16", 768, 384, 3, 1, 1, 16),
    ("384->384 k3 @8 gsc (x2)", 384, 384, 3, 1, 1, 8), ("384->768 k3 @8", 384, 768, 3, 1, 1, 8), ("768->768 k3 @8", 768, 768, 3, 1, 1, 8),
]
B = 2
tot = [0, 0, 0]
for name, ci, co, k, s, p, S in cfgs:
    x = torch.randn(B, ci, S, S, S, device=dev, dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn(co, ci, k, k, k, device=dev, dtype=torch.bfloat16, requires_grad=True) * 0.05
    y = F.conv3d(x, w, None, s, p)
    g = torch.randn_like(y)
    flops = 2 * B * co * ci * k**3 * y.shape[2] * y.shape[3] * y.shape[4]
    try:
        t_f = tm(lambda: F.conv3d(x, w, None, s, p))
        t_d = tm(lambda: torch.autograd.grad(F.conv3d(x, w.detach(), None, s, p), x, g)) - t_f
        t_w = tm(lambda: torch.autograd.grad(F.conv3d(x.detach(), w, None, s, p), w, g)) - t_f
    except Exception as e:
        print(name, "ERR", str(e)[:100]); continue
    print(f"{name:34s} GF={flops/1e9:7.1f} fwd {t_f:8.2f} ms ({flops/t_f/1e9:6.1f} TF/s)  dgrad {t_d:8.2f} ms  wgrad {t_w:8.2f} ms", flush=True)
    del x, w, y, g
    torch.cuda.empty_cache()
