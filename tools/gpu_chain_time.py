"""A/B timings on the GPU box: conv3d forward default vs chained-K-parts kernel, fused clip+SGD / cross entropy vs ATen."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmamba_amd import lib as L, ops_raw, train_ops

dev = "cuda:0"
hip = L.get_lib()


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for (B, cin, cout, S) in [(2, 48, 48, 128), (2, 48, 48, 64), (2, 96, 96, 64), (2, 192, 192, 32), (2, 96, 48, 128)]:
    x = torch.randn(B, cin, S, S, S, device=dev).bfloat16()
    w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device=dev)).bfloat16()
    wps = [ops_raw.pack_conv3d_weight(w[:, i:i + 48]) for i in range(0, cin, 48)]

    def run(**kw):
        out = None
        for i, wp in enumerate(wps):
            out = ops_raw.conv3d_k3_fwd(hip, x[:, 48 * i:48 * i + 48], wp, None, out=out, accumulate=i > 0, **kw)
        return out
    try:
        fl = 2.0 * B * S ** 3 * cin * cout * 27
        base = run()
        line = f"conv fwd {cin}->{cout} @{S}^3 B={B}:"
        for name, kw in (("reduce", {}), ("chain", dict(chain=True)), ("chain48", dict(chain=True, pitch48=True)), ("chain32", dict(chain32=True))):
            t = timeit(lambda: run(**kw))
            d = (run(**kw).float() - base.float()).abs().max().item()
            line += f"  {name} {t:.3f} ms ({fl / t / 1e9:.0f} TF/s, maxdiff {d:.2g})"
        print(line, flush=True)
    except RuntimeError as e:
        print(f"conv fwd {cin}->{cout} @{S}^3: {e}", flush=True)

if os.environ.get("SEGM_TIME_CONV_ONLY"):
    sys.exit(0)

# optimizer: 67 M fp32 parameters in ~290 tensors (SegMamba-like size mix)
sizes = [768 * 768 * 27] * 2 + [384 * 384 * 27] * 6 + [192 * 192 * 27] * 8 + [96 * 96 * 27] * 10 + [48 * 48 * 27] * 12 + [384 * 768] * 8 + [96] * 120 + [768] * 120
params = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in sizes]
grads = [torch.randn(n, device=dev) for n in sizes]
for p, g in zip(params, grads):
    p.grad = g
opt = torch.optim.SGD(params, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True)
fopt = train_ops.FusedClipSGD(params, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True, max_norm=12.0)


def aten_step():
    torch.nn.utils.clip_grad_norm_(params, 12.0)
    opt.step()


print(f"optimizer ({sum(sizes) / 1e6:.1f} M params, {len(sizes)} tensors): ATen clip+SGD {timeit(aten_step):.3f} ms  fused {timeit(fopt.step):.3f} ms", flush=True)
t = time.perf_counter(); fopt.step(); print(f"  fused host time per call {1e3 * (time.perf_counter() - t):.3f} ms")

logits = torch.randn(2, 4, 128, 128, 128, device=dev).bfloat16().requires_grad_()
labels = torch.randint(0, 4, (2, 128, 128, 128), device=dev)


def aten_ce():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l = torch.nn.functional.cross_entropy(logits, labels)
    l.backward()


def fused_ce():
    train_ops.cross_entropy(logits, labels).backward()


print(f"cross entropy fwd+bwd (2,4,128^3) bf16: ATen {timeit(aten_ce):.3f} ms  fused {timeit(fused_ce):.3f} ms", flush=True)
