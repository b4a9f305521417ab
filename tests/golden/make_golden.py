"""Generate the golden fixtures in this directory from the REFERENCE's own code.

Run in the build container only (needs /root/reference, CPU):
    python tests/golden/make_golden.py

The reference's native extensions cannot be built here (CUDA), but its pure-PyTorch
definitions can be imported once the two extension modules are stubbed
(SURVEY.md §8c).  Everything below calls reference functions *as they are*; where the
reference routes through a CUDA op (`selective_scan_fn`, `causal_conv1d_fn`,
`mamba_inner_fn_no_out_proj`) the name is rebound to the reference's own `*_ref`
function, never to code from this repository.

Fixtures (all float32 unless noted, seeds/distributions follow the reference tests,
mamba/tests/ops/test_selective_scan.py:53-88, causal-conv1d/tests/test_causal_conv1d.py:36-55):
  scan_L{64,256}_G{1,2}.npz      selective_scan_ref outputs + all gradients
  conv1d_w{2,3,4}.npz            causal_conv1d_ref outputs + gradients (silu, bias)
  inner_no_out_proj.npz          mamba_inner_ref with an identity out-projection
  mamba_v3.npz                   reference Mamba(bimamba_type="v3").forward, CPU ref ops
  segmamba_tiny.npz              reference SegMamba forward, 32^3 input, tiny widths
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    for name in ("causal_conv1d_cuda", "selective_scan_cuda"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REF, "causal-conv1d"))
    sys.path.insert(0, REF)                      # vendored monai, model_segmamba
    import causal_conv1d.causal_conv1d_interface as cci
    pkg = types.ModuleType("mamba_ssm")
    pkg.__path__ = [os.path.join(REF, "mamba", "mamba_ssm")]
    sys.modules["mamba_ssm"] = pkg               # skip mamba_ssm/__init__.py (LM imports fail)
    import mamba_ssm.ops.selective_scan_interface as ssi
    # CUDA-only entry points -> the reference's own pure-PyTorch definitions
    ssi.causal_conv1d_fn = cci.causal_conv1d_ref
    ssi.selective_scan_fn = ssi.selective_scan_ref

    def no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, B=None, C=None, D=None, delta_bias=None,
                    B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
        eye = torch.eye(xz.shape[1] // 2, dtype=xz.dtype)
        y = ssi.mamba_inner_ref(xz, conv_w, conv_b, x_proj_w, dt_proj_w, eye, None, A, B, C, D,
                                delta_bias, B_proj_bias, C_proj_bias, delta_softplus)
        return y.transpose(1, 2)                 # (b, l, d) -> (b, d, l), what the fused fn returns
    ssi.mamba_inner_fn_no_out_proj = no_out_proj
    import mamba_ssm.modules.mamba_simple as ms
    ms.mamba_inner_fn_no_out_proj = no_out_proj
    pkg.Mamba = ms.Mamba
    return cci, ssi, ms


def named_fill(state_dict, scale=0.5):
    """Deterministic weights that depend only on the key name and shape (shared with the tests)."""
    out = {}
    for k, v in state_dict.items():
        seed = int.from_bytes(hashlib.sha256(k.encode()).digest()[:4], "little")
        g = torch.Generator().manual_seed(seed)
        fan = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 1
        t = torch.randn(v.shape, generator=g, dtype=torch.float32) * (scale / np.sqrt(fan))
        if k.endswith("A_log") or "_log" in k:
            t = torch.log(torch.rand(v.shape, generator=g) * 4 + 0.25)
        elif k.endswith(".D") or ".D_" in k:
            t = torch.randn(v.shape, generator=g)
        elif "dt_proj" in k and k.endswith("bias"):
            t = torch.rand(v.shape, generator=g) * 2 - 3.0
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        out[k] = t.to(v.dtype)
    return out


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else v)
                                                       for k, v in arrs.items() if v is not None})
    print("wrote", name, {k: tuple(v.shape) for k, v in arrs.items() if torch.is_tensor(v)})


def gen_scan(ssi):
    for L in (64, 256):
        for G in (1, 2):
            torch.manual_seed(0)
            b, dim, N = 2, 4, 8
            A = (-0.5 * torch.rand(dim, N)).requires_grad_()
            shape = (b, N, L) if G == 1 else (b, G, N, L)
            B = torch.randn(*shape, requires_grad=True)
            C = torch.randn(*shape, requires_grad=True)
            D = torch.randn(dim, requires_grad=True)
            z = torch.randn(b, dim, L, requires_grad=True)
            db = (0.5 * torch.rand(dim)).requires_grad_()
            u = torch.randn(b, dim, L, requires_grad=True)
            delta = (0.5 * torch.rand(b, dim, L)).requires_grad_()
            out, last = ssi.selective_scan_ref(u, delta, A, B, C, D, z=z, delta_bias=db, delta_softplus=True,
                                               return_last_state=True)
            g = torch.randn_like(out)
            out.backward(g)
            save(f"scan_L{L}_G{G}.npz", u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, delta_bias=db, g=g,
                 out=out, last_state=last, du=u.grad, ddelta=delta.grad, dA=A.grad, dB=B.grad, dC=C.grad,
                 dD=D.grad, dz=z.grad, ddelta_bias=db.grad)


def gen_conv(cci):
    for width in (2, 3, 4):
        torch.manual_seed(0)
        b, dim, L = 2, 24, 151
        x = torch.randn(b, dim, L, requires_grad=True)
        w = torch.randn(dim, width, requires_grad=True)
        bias = torch.randn(dim, requires_grad=True)
        out = cci.causal_conv1d_ref(x, w, bias, activation="silu")
        g = torch.randn_like(out)
        out.backward(g)
        out_lin = cci.causal_conv1d_ref(x.detach(), w.detach(), None, activation=None)
        save(f"conv1d_w{width}.npz", x=x, weight=w, bias=bias, g=g, out=out, out_nobias_noact=out_lin,
             dx=x.grad, dweight=w.grad, dbias=bias.grad)


def gen_inner(ssi):
    torch.manual_seed(0)
    b, d_model, L, N = 2, 24, 96, 16
    dim, R = 2 * d_model, 2
    xz = torch.randn(b, 2 * dim, L, requires_grad=True)
    conv_w = torch.randn(dim, 1, 4, requires_grad=True)
    conv_b = torch.randn(dim, requires_grad=True)
    x_proj_w = (torch.randn(R + 2 * N, dim) / dim ** 0.5).requires_grad_()
    dt_proj_w = (torch.randn(dim, R) / R ** 0.5).requires_grad_()
    A = (-torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(dim, 1))).requires_grad_()
    D = torch.ones(dim, requires_grad=True)
    db = (torch.rand(dim) - 4.0).requires_grad_()
    out = ssi.mamba_inner_fn_no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, None, None, D,
                                         delta_bias=db, delta_softplus=True)
    g = torch.randn_like(out)
    out.backward(g)
    save("inner_no_out_proj.npz", xz=xz, conv_w=conv_w, conv_b=conv_b, x_proj_w=x_proj_w, dt_proj_w=dt_proj_w,
         A=A, D=D, delta_bias=db, g=g, out=out, dxz=xz.grad, dconv_w=conv_w.grad, dconv_b=conv_b.grad,
         dx_proj_w=x_proj_w.grad, ddt_proj_w=dt_proj_w.grad, dA=A.grad, dD=D.grad, ddelta_bias=db.grad)


def gen_mamba(ms):
    torch.manual_seed(0)
    d_model, L, ns, b = 16, 64, 8, 2
    m = ms.Mamba(d_model=d_model, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=ns)
    m.load_state_dict(named_fill(m.state_dict()))
    x = torch.randn(b, L, d_model, requires_grad=True)
    y = m(x)
    g = torch.randn_like(y)
    y.backward(g)
    grads = {"grad__" + k: p.grad for k, p in m.named_parameters()}
    save("mamba_v3.npz", x=x, g=g, y=y, dx=x.grad, nslices=np.array(ns), **grads)


def gen_segmamba():
    from model_segmamba.segmamba import SegMamba
    torch.manual_seed(0)
    m = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 8, 16, 32], hidden_size=32)  # reference hard-codes 48 into the head (segmamba.py:317)
    m.load_state_dict(named_fill(m.state_dict()))
    m.eval()
    x = torch.rand(1, 4, 32, 32, 32, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = m(x)
    save("segmamba_tiny.npz", x_seed=np.array(1234), y_sub=y[:, :, ::2, ::2, ::2].contiguous(),
         y_mean=y.mean(), y_std=y.std(), y_abs_sum=y.abs().sum(), nkeys=np.array(len(m.state_dict())))
    # key list + shapes of the full-size model (the checkpoint-compatibility contract)
    full = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384])
    with open(os.path.join(HERE, "segmamba_state_dict_keys.txt"), "w") as f:
        for k, v in full.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")


if __name__ == "__main__":
    cci, ssi, ms = load_reference()
    gen_scan(ssi)
    gen_conv(cci)
    gen_inner(ssi)
    gen_mamba(ms)
    gen_segmamba()
