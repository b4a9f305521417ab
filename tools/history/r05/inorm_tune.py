"""Round 5: the InstanceNorm streaming passes against the ~6.3 TB/s a float4 copy reaches on this part (MI355X_MICROARCH.md):
workgroup count (SEGM_NORM_WGS), packets in flight (SEGM_NORM_DEEP) and non-temporal accesses (SEGM_NORM_NT) at the three largest
layer shapes, on padded volumes as in the step.  GB/s = algorithmic bytes (every tensor once per pass that touches it) / time."""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmamba_amd import lib as L, ops_raw
hip = L.get_lib()


def t(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


def vol(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = ops_raw.volume_empty(shape[0], shape[1], shape[2:], torch.bfloat16, "cuda")
    v.copy_(torch.randn(shape, device="cuda", generator=g))
    return v


for shape in ((2, 48, 128, 128, 128), (2, 96, 64, 64, 64), (2, 192, 32, 32, 32)):
    x, res, dy = vol(shape, 1), vol(shape, 2), vol(shape, 3)
    nb = x.numel() * 2
    os.environ.pop("SEGM_NORM_WGS", None); os.environ.pop("SEGM_NORM_NT", None); os.environ.pop("SEGM_NORM_DEEP", None)
    y, mean, rstd = ops_raw.instnorm_fwd(hip, x, res, "leaky_relu")
    y0, m0, r0 = ops_raw.instnorm_fwd(hip, x, None, "leaky_relu")
    dx0, _ = ops_raw.instnorm_bwd(hip, x, dy, m0, r0, None, "leaky_relu")
    print(f"== {shape}, {nb / 1e6:.0f} MB per tensor", flush=True)
    for wgs, deep, nt in itertools.product((int(v) for v in os.environ.get('TUNE_WGS', '1024,2048,4096,8192,16384').split(',')), (0, 1), (int(v) for v in os.environ.get('TUNE_NT', '0,1').split(','))):
        os.environ["SEGM_NORM_WGS"], os.environ["SEGM_NORM_DEEP"], os.environ["SEGM_NORM_NT"] = str(wgs), str(deep), str(nt)
        f = t(lambda: ops_raw.instnorm_fwd(hip, x, None, "leaky_relu"))            # stats: read x; apply: read x, write y
        fr = t(lambda: ops_raw.instnorm_fwd(hip, x, res, "leaky_relu"))           # + read residual
        b = t(lambda: ops_raw.instnorm_bwd(hip, x, dy, m0, r0, None, "leaky_relu"))                       # stats: x, dy; apply: x, dy -> dx
        br = t(lambda: ops_raw.instnorm_bwd(hip, x, dy, mean, rstd, y, "leaky_relu", want_dresidual=True))  # stats: x, dy, y -> g; apply: x, g -> dx
        y1, m1, r1 = ops_raw.instnorm_fwd(hip, x, None, "leaky_relu")
        dx1, _ = ops_raw.instnorm_bwd(hip, x, dy, m0, r0, None, "leaky_relu")
        ok = torch.equal(y1, y0) and float((dx1.float() - dx0.float()).abs().max()) <= 2.0 ** -7 * float(dx0.float().abs().max())
        print(f"wgs {wgs:6d} deep {deep} nt {nt}: fwd {f:.3f} ms {3 * nb / f * 1e-6:5.0f} GB/s | fwd+res {fr:.3f} {4 * nb / fr * 1e-6:5.0f} | "
              f"bwd {b:.3f} {5 * nb / b * 1e-6:5.0f} | bwd+res {br:.3f} {7 * nb / br * 1e-6:5.0f} | same {ok}", flush=True)
