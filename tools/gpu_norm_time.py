"""GPU timing of the InstanceNorm(+LeakyReLU) kernels against the ATen chain at the stage-0 shape."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for shape in ((2, 48, 128, 128, 128), (2, 96, 64, 64, 64), (2, 384, 16, 16, 16)):
    x = torch.randn(shape, device="cuda").bfloat16()
    res = torch.randn(shape, device="cuda").bfloat16()
    dy = torch.randn(shape, device="cuda").bfloat16()
    nbytes = x.numel() * 2
    f = t(lambda: ops_raw.instnorm_fwd(hip, x, None, "leaky_relu"))
    fr = t(lambda: ops_raw.instnorm_fwd(hip, x, res, "leaky_relu"))
    y, mean, rstd = ops_raw.instnorm_fwd(hip, x, res, "leaky_relu")
    b = t(lambda: ops_raw.instnorm_bwd(hip, x, dy, mean, rstd, None, "leaky_relu"))
    br = t(lambda: ops_raw.instnorm_bwd(hip, x, dy, mean, rstd, y, "leaky_relu", want_dresidual=True))
    fa = t(lambda: F.leaky_relu(F.instance_norm(x), 0.01))
    xr = x.clone().requires_grad_()
    ya = F.leaky_relu(F.instance_norm(xr), 0.01)
    ba = t(lambda: torch.autograd.grad(ya, xr, dy, retain_graph=True))
    print(f"{shape}: fwd {f:.3f} ms ({3 * nbytes / f * 1e-6:.0f} GB/s)  fwd+res {fr:.3f} ms ({4 * nbytes / fr * 1e-6:.0f} GB/s)  "
          f"bwd {b:.3f} ms ({5 * nbytes / b * 1e-6:.0f} GB/s)  bwd+res {br:.3f} ms ({7 * nbytes / br * 1e-6:.0f} GB/s)  "
          f"| ATen fwd {fa:.3f} ms bwd {ba:.3f} ms", flush=True)
