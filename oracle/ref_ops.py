"""CPU ORACLE (test infrastructure, never a product path).

Plain-PyTorch restatement of the reference's own pure-PyTorch definitions of
the hot-path operators.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; the product path
(`segmamba_amd`) must never do so and fails loudly without its HIP library.

Pinning: `tests/golden/make_golden.py` imports the *reference's* functions
from /root/reference (in the build container) and stores their outputs and
gradients for seeded inputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those
fixtures (fp32, max abs err <= 2e-6 relative to output scale), so this file
is pinned to the reference's behaviour, not merely to its documentation.

Each function cites the reference lines it restates
(paths relative to the reference root).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# selective scan
# --------------------------------------------------------------------------------------
def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False, compute_dtype=torch.float32):
    """mamba/mamba_ssm/ops/selective_scan_interface.py:86-152 (`selective_scan_ref`),
    real `A`; `B` / `C` input-dependent (the SegMamba path) or constant per channel (:104-110, :122-133 - the
    `is_variable_B/C = False` rows of the reference's test matrix).

    u, delta, z : (batch, dim, L)            A : (dim, N)      D, delta_bias : (dim,)
    B, C        : (batch, N, L) or (batch, G, N, L) with dim % G == 0, or constant (dim, N)
    Returns out (batch, dim, L) in u.dtype [, last_state (batch, dim, N) in compute_dtype].

    h_t = exp(delta_t * A) * h_{t-1} + delta_t * u_t * B_t ; y_t = <C_t, h_t> + D*u_t ; out = y * silu(z)

    The reference materialises (batch, dim, L, N) tensors; this walks t and keeps only the
    (batch, dim, N) state, which is the same arithmetic in the same order per step.
    `compute_dtype=torch.float64` gives a higher-precision variant for tolerance studies.
    """
    if A.is_complex():
        raise NotImplementedError("complex A is not on the SegMamba path (SURVEY.md §2.1)")
    dtype_in = u.dtype
    cd = compute_dtype
    u_ = u.to(cd)
    dl = delta.to(cd)
    if delta_bias is not None:
        dl = dl + delta_bias.to(cd)[..., None]
    if delta_softplus:
        dl = F.softplus(dl)
    batch, dim, L = u_.shape
    N = A.shape[1]
    A_ = A.to(cd)
    def per_channel(M):
        M = M.to(cd)
        if M.dim() == 2:                                   # constant (dim, N): the same row at every step (:122-123, :131-132)
            return M[None, :, :, None].expand(batch, dim, N, L)
        if M.dim() == 3:
            M = M[:, None]
        # (batch, G, N, L) -> (batch, dim, N, L) by repeating each group over its dim//G channels
        return M.repeat_interleave(dim // M.shape[1], dim=1)
    Bm = per_channel(B)
    Cm = per_channel(C)

    h = u_.new_zeros((batch, dim, N))
    ys = []
    for t in range(L):
        dA = torch.exp(dl[:, :, t, None] * A_)                               # (b, d, n)
        h = dA * h + (dl[:, :, t] * u_[:, :, t])[..., None] * Bm[:, :, :, t]
        ys.append((h * Cm[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u_ * D.to(cd)[:, None]
    if z is not None:
        out = out * F.silu(z.to(cd))
    out = out.to(dtype_in)
    return (out, h) if return_last_state else out


# --------------------------------------------------------------------------------------
# causal depthwise conv1d
# --------------------------------------------------------------------------------------
def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """causal-conv1d/causal_conv1d/causal_conv1d_interface.py:49-65 (`causal_conv1d_ref`).

    x : (batch, dim, L)   weight : (dim, width)   bias : (dim,)
    out[b, d, t] = bias[d] + sum_w weight[d, w] * x[b, d, t - (width-1-w)]  (zero left pad), then SiLU.
    """
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    dtype_in = x.dtype
    xw = x.to(weight.dtype)
    L = xw.shape[-1]
    dim, width = weight.shape
    out = F.conv1d(xw, weight[:, None, :], bias, padding=width - 1, groups=dim)[..., :L]
    if activation is not None:
        out = F.silu(out)
    return out.to(dtype_in)


# --------------------------------------------------------------------------------------
# fused inner function (conv1d -> x_proj -> dt_proj -> scan), no output projection
# --------------------------------------------------------------------------------------
def mamba_inner_no_out_proj_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                A, D=None, delta_bias=None, delta_softplus=True):
    """mamba/mamba_ssm/ops/selective_scan_interface.py:636-670 (`mamba_inner_ref`) without the final
    `F.linear(..., out_proj_weight)`, i.e. the value `MambaInnerFnNoOutProj` returns (:155-224).

    xz : (batch, 2*dim, L) -> out_z : (batch, dim, L)
    """
    L = xz.shape[-1]
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    xc = causal_conv1d_ref(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias, "silu")
    batch, dim, _ = xc.shape
    x_dbl = F.linear(xc.permute(0, 2, 1).reshape(batch * L, dim), x_proj_weight)       # (b*l, R+2N)
    delta = (delta_proj_weight @ x_dbl[:, :R].t()).reshape(dim, batch, L).permute(1, 0, 2)
    Bm = x_dbl[:, R:R + N].reshape(batch, L, N).permute(0, 2, 1).contiguous()
    Cm = x_dbl[:, -N:].reshape(batch, L, N).permute(0, 2, 1).contiguous()
    return selective_scan_ref(xc, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias,
                              delta_softplus=delta_softplus)


def _inner_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, B_proj_bias, C_proj_bias):
    """the shared front of `mamba_inner_ref` / `bimamba_inner_ref` (:642-667 = :679-704): conv1d + SiLU, x_proj, dt_proj and
    the B / C the scan sees - columns of x_dbl where they are input-dependent (None), the given (dim, N) constants otherwise"""
    L = xz.shape[-1]
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    xc = causal_conv1d_ref(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias, "silu")
    batch, dim, _ = xc.shape
    x_dbl = F.linear(xc.permute(0, 2, 1).reshape(batch * L, dim), x_proj_weight)
    delta = (delta_proj_weight @ x_dbl[:, :R].t()).reshape(dim, batch, L).permute(1, 0, 2)
    if B is None:
        B = x_dbl[:, R:R + N]
        if B_proj_bias is not None:
            B = B + B_proj_bias.to(B.dtype)
        B = B.reshape(batch, L, N).permute(0, 2, 1).contiguous()
    if C is None:
        C = x_dbl[:, -N:]
        if C_proj_bias is not None:
            C = C + C_proj_bias.to(C.dtype)
        C = C.reshape(batch, L, N).permute(0, 2, 1).contiguous()
    return xc, z, delta, B, C


def mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                    A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """mamba/mamba_ssm/ops/selective_scan_interface.py:636-670 (`mamba_inner_ref`): real A, B / C None (input-dependent) or
    constant (dim, N).  xz (batch, 2*dim, L) -> (batch, L, out_features)."""
    xc, z, delta, B, C = _inner_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                                            B_proj_bias, C_proj_bias)
    y = selective_scan_ref(xc, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    return F.linear(y.permute(0, 2, 1), out_proj_weight, out_proj_bias)


def bimamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                      A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                      delta_softplus=True):
    """mamba/mamba_ssm/ops/selective_scan_interface.py:673-709 (`bimamba_inner_ref`): the forward scan with A plus a scan of
    the time-reversed conv output / delta / B / C / z with A_b (every other weight shared), flipped back and added."""
    xc, z, delta, B, C = _inner_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                                            B_proj_bias, C_proj_bias)
    y = selective_scan_ref(xc, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    y_b = selective_scan_ref(xc.flip([-1]), delta.flip([-1]), A_b, B.flip([-1]), C.flip([-1]), D, z.flip([-1]),
                             delta_bias, delta_softplus=True)
    return F.linear((y + y_b.flip([-1])).permute(0, 2, 1), out_proj_weight, out_proj_bias)


# --------------------------------------------------------------------------------------
# Mamba block, bimamba_type == "v3"
# --------------------------------------------------------------------------------------
def slice_interleave(x, nslices):
    """mamba_simple.py:245-247: new[..., j*ns + k] = old[..., k*(L/ns) + j]."""
    L = x.shape[-1]
    return x.reshape(*x.shape[:-1], nslices, L // nslices).transpose(-1, -2).reshape(*x.shape[:-1], L)


def slice_deinterleave(x, nslices):
    """mamba_simple.py:261: inverse of `slice_interleave`."""
    L = x.shape[-1]
    return x.reshape(*x.shape[:-1], L // nslices, nslices).transpose(-1, -2).reshape(*x.shape[:-1], L)


def mamba_v3_forward_ref(hidden_states, p, nslices):
    """mamba/mamba_ssm/modules/mamba_simple.py:188-264, `bimamba_type == "v3"` branch.

    hidden_states : (batch, L, d_model); `p` maps the reference parameter names
    (`in_proj.weight`, `conv1d.weight`, ..., `A_log`, `D`, `*_b`, `*_s`, `out_proj.weight`) to tensors.
    """
    batch, L, _ = hidden_states.shape
    xz = (p["in_proj.weight"] @ hidden_states.reshape(batch * L, -1).t())           # (2D, b*l)
    xz = xz.reshape(-1, batch, L).permute(1, 0, 2)                                    # (b, 2D, l)
    if p.get("in_proj.bias") is not None:
        xz = xz + p["in_proj.bias"].to(xz.dtype)[:, None]

    def direction(suffix, xz_dir):
        A = -torch.exp(p["A" + suffix + "_log"].float())
        return mamba_inner_no_out_proj_ref(
            xz_dir, p["conv1d" + suffix + ".weight"], p["conv1d" + suffix + ".bias"],
            p["x_proj" + suffix + ".weight"], p["dt_proj" + suffix + ".weight"], A,
            p["D" + suffix].float(), delta_bias=p["dt_proj" + suffix + ".bias"].float(), delta_softplus=True)

    out = direction("", xz)
    out_b = direction("_b", xz.flip([-1])).flip([-1])
    out_s = slice_deinterleave(direction("_s", slice_interleave(xz, nslices)), nslices)
    y = (out + out_b + out_s).permute(0, 2, 1)                                        # (b, l, D)
    return F.linear(y, p["out_proj.weight"], p.get("out_proj.bias"))


# --------------------------------------------------------------------------------------
# single-token decode
# --------------------------------------------------------------------------------------
def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None):
    """causal-conv1d/causal_conv1d/causal_conv1d_interface.py:84-104.  x (batch, dim); conv_state (batch, dim, width),
    updated in place; weight (dim, width); bias (dim)."""
    dtype_in = x.dtype
    conv_state.copy_(torch.roll(conv_state, shifts=-1, dims=-1))
    conv_state[:, :, -1] = x
    out = torch.sum(conv_state * weight, dim=-1)
    if bias is not None:
        out = out + bias
    return (out if activation is None else F.silu(out)).to(dtype=dtype_in)


def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """mamba/mamba_ssm/ops/triton/selective_state_update.py:157-192.  state (batch, dim, dstate), updated in place."""
    if dt_bias is not None:
        dt = dt + dt_bias
    dt = F.softplus(dt) if dt_softplus else dt
    dA = torch.exp(dt[:, :, None] * A)
    dB = dt[:, :, None] * B[:, None, :]
    state.copy_(state * dA + dB * x[:, :, None])
    out = torch.einsum("bdn,bn->bd", state.to(C.dtype), C)
    if D is not None:
        out = out + (x * D).to(out.dtype)
    return (out if z is None else out * F.silu(z)).to(x.dtype)


def mamba_step_ref(hidden_states, p, conv_state, ssm_state):
    """mamba/mamba_ssm/modules/mamba_simple.py:356-401 (`Mamba.step`): one token through the forward-direction
    parameters; hidden_states (batch, 1, d_model); the two states are updated in place."""
    d_state = p["A_log"].shape[1]
    dt_rank = p["dt_proj.weight"].shape[1]
    xz = F.linear(hidden_states.squeeze(1), p["in_proj.weight"], p.get("in_proj.bias"))
    x, z = xz.chunk(2, dim=-1)
    x = causal_conv1d_update_ref(x, conv_state, p["conv1d.weight"].squeeze(1), p["conv1d.bias"], "silu")
    x_db = F.linear(x, p["x_proj.weight"])
    dt, B, C = torch.split(x_db, [dt_rank, d_state, d_state], dim=-1)
    dt = F.linear(dt, p["dt_proj.weight"])
    A = -torch.exp(p["A_log"].float())
    y = selective_state_update_ref(ssm_state, x, dt, A, B, C, p["D"], z=z, dt_bias=p["dt_proj.bias"], dt_softplus=True)
    return F.linear(y, p["out_proj.weight"], p.get("out_proj.bias")).unsqueeze(1)


def mamba_prefill_ref(hidden_states, p, conv_state, ssm_state):
    """mamba/mamba_ssm/modules/mamba_simple.py:196-208, 265-355: `Mamba.forward` with `inference_params` at
    seqlen_offset 0 - the uni-directional branch on the forward-direction parameters, leaving the last d_conv inputs
    of the convolution in conv_state and the final SSM state in ssm_state."""
    batch, L, _ = hidden_states.shape
    d_state = p["A_log"].shape[1]
    dt_rank = p["dt_proj.weight"].shape[1]
    d_conv = p["conv1d.weight"].shape[-1]
    xz = (p["in_proj.weight"] @ hidden_states.reshape(batch * L, -1).t()).reshape(-1, batch, L).permute(1, 0, 2)
    if p.get("in_proj.bias") is not None:
        xz = xz + p["in_proj.bias"].to(xz.dtype)[:, None]
    x, z = xz.chunk(2, dim=1)
    conv_state.copy_(x[:, :, -d_conv:])
    x = causal_conv1d_ref(x, p["conv1d.weight"].squeeze(1), p["conv1d.bias"], "silu")
    x_dbl = F.linear(x.permute(0, 2, 1).reshape(batch * L, -1), p["x_proj.weight"])
    dt, B, C = torch.split(x_dbl, [dt_rank, d_state, d_state], dim=-1)
    dt = (p["dt_proj.weight"] @ dt.t()).reshape(-1, batch, L).permute(1, 0, 2)
    B = B.reshape(batch, L, d_state).permute(0, 2, 1).contiguous()
    C = C.reshape(batch, L, d_state).permute(0, 2, 1).contiguous()
    A = -torch.exp(p["A_log"].float())
    y, last = selective_scan_ref(x, dt, A, B, C, p["D"].float(), z=z, delta_bias=p["dt_proj.bias"].float(),
                                 delta_softplus=True, return_last_state=True)
    ssm_state.copy_(last)
    return F.linear(y.permute(0, 2, 1), p["out_proj.weight"], p.get("out_proj.bias"))


# --------------------------------------------------------------------------------------
# closed-form backward of the scan (Appendix A of SURVEY.md), fp64, used as an independent check
# --------------------------------------------------------------------------------------
def selective_scan_bwd_closed_form(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=True):
    """Gradients from the explicit recurrences the native kernels implement
    (mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:161-478), evaluated in fp64 with
    G == 1.  Returns dict(du, ddelta, dA, dB, dC, dD, dz, ddelta_bias).
    """
    f8 = torch.float64
    u, delta, A, B, C, dout = (t.to(f8) for t in (u, delta, A, B, C, dout))
    if B.dim() == 4:
        B = B[:, 0]
    if C.dim() == 4:
        C = C[:, 0]
    batch, dim, L = u.shape
    N = A.shape[1]
    raw = delta + (delta_bias.to(f8)[:, None] if delta_bias is not None else 0.0)
    dl = F.softplus(raw) if delta_softplus else raw
    hs = torch.zeros(batch, dim, L + 1, N, dtype=f8)
    for t in range(L):
        a = torch.exp(dl[:, :, t, None] * A)
        hs[:, :, t + 1] = a * hs[:, :, t] + (dl[:, :, t] * u[:, :, t])[..., None] * B[:, None, :, t]
    y = torch.einsum("bdtn,bnt->bdt", hs[:, :, 1:], C) + (D.to(f8)[:, None] * u if D is not None else 0.0)
    if z is not None:
        zz = z.to(f8)
        sig = torch.sigmoid(zz)
        g = dout * zz * sig
        dz = dout * y * sig * (1 + zz * (1 - sig))
    else:
        g, dz = dout, None
    du = (D.to(f8)[:, None] * g) if D is not None else torch.zeros_like(u)
    ddl = torch.zeros_like(dl)
    dA = torch.zeros_like(A)
    dB = torch.zeros_like(B)
    dC = torch.einsum("bdt,bdtn->bnt", g, hs[:, :, 1:])
    dh_next = torch.zeros(batch, dim, N, dtype=f8)     # a_{t+1} * dh_{t+1}
    for t in range(L - 1, -1, -1):
        a = torch.exp(dl[:, :, t, None] * A)
        dh = dh_next + g[:, :, t, None] * C[:, None, :, t]
        hprev = hs[:, :, t]
        da = dh * hprev                                 # dL/da_{n,t}
        du[:, :, t] += dl[:, :, t] * (dh * B[:, None, :, t]).sum(-1)
        ddl[:, :, t] = (dh * B[:, None, :, t]).sum(-1) * u[:, :, t] + (da * a * A).sum(-1)
        dA += (da * a * dl[:, :, t, None]).sum(0)
        dB[:, :, t] = (dh * (dl[:, :, t] * u[:, :, t])[..., None]).sum(1)
        dh_next = a * dh
    if delta_softplus:
        ddelta = ddl * torch.sigmoid(raw)
    else:
        ddelta = ddl
    out = dict(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC, dz=dz,
               dD=(g * u).sum((0, 2)) if D is not None else None,
               ddelta_bias=ddelta.sum((0, 2)) if delta_bias is not None else None)
    return out


def softplus_inverse(dt):
    """mamba_simple.py:105 : inv_dt = dt + log(-expm1(-dt))."""
    return dt + torch.log(-torch.expm1(-dt))


def algorithmic_bytes_scan(batch, dim, L, N, esize, G=1, backward=False):
    """the byte count `roofline.achieved` is computed from."""
    if not backward:
        return esize * batch * L * (5 * dim + 2 * G * N)
    return batch * L * (esize * (8 * dim + 2 * G * N) + 8 * G * N)
