"""`mamba_ssm.ops.selective_scan_interface` on the MI355X kernels.

Host-side mirror of reference mamba/mamba_ssm/ops/selective_scan_interface.py:
  SelectiveScanFn / selective_scan_fn          :14-83
  MambaInnerFnNoOutProj / mamba_inner_fn_no_out_proj   :155-289, :627-633   (the one SegMamba uses)
  mamba_inner_fn / bimamba_inner_fn            :292-624 (same inner pipeline + output projection)
Same names, argument order, layouts ((batch, dim, seqlen) tensors, B/C as (batch, [groups,] dstate,
seqlen)) and error behaviour.  What changes underneath:

  * the native ops are `segm_*` entry points of libsegmamba_hip.so (no fallback if it is missing);
  * the inner pipeline exists once (`MambaInnerCore`) and is layout- and time-order-generic: the reference
    entry points call it with channel-first views, `Mamba.forward` calls it on the channel-last (B, L, 2D)
    tensor the in-projection GEMM naturally produces, with the reversed / slice-interleaved directions as
    an index map inside the kernels instead of `flip` / `stack` / `permute` copies
    (reference mamba_simple.py:231,245-247,261);
  * B and C are never transposed into separate contiguous tensors (reference :193,205): the scan reads
    them as strided views of `x_dbl`, and dB / dC are accumulated straight into the matching columns of
    the fp32 `dx_dbl` buffer (reference :257-268 does two more transposing copies).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import lib as L
from . import ops_raw
from .linear import tn_matmul


def _custom_fwd(fn):
    return torch.amp.custom_fwd(fn, device_type="cuda")


def _custom_bwd(fn):
    return torch.amp.custom_bwd(fn, device_type="cuda")


# ---------------------------------------------------------------------------------------------------------
# standalone selective scan
# ---------------------------------------------------------------------------------------------------------
class SelectiveScanFn(torch.autograd.Function):
    """reference: selective_scan_interface.py:14-73"""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        if u.stride(-1) != 1:
            u = u.contiguous()
        if delta.stride(-1) != 1:
            delta = delta.contiguous()
        if D is not None:
            D = D.contiguous()
        if B.stride(-1) != 1:
            B = B.contiguous()
        if C.stride(-1) != 1:
            C = C.contiguous()
        if z is not None and z.stride(-1) != 1:
            z = z.contiguous()
        if A.is_complex() or B.dim() < 3 or C.dim() < 3:
            raise RuntimeError("only real A with input-dependent B and C is supported (the SegMamba path)")
        A32 = A.float().contiguous()
        D32 = D.float() if D is not None else None
        db32 = delta_bias.float().contiguous() if delta_bias is not None else None
        r = ops_raw.scan_fwd(L.get_lib(), u, delta, A32, B, C, D32, z, db32, delta_softplus, channel_last=False,
                             need_out=True, need_ckpt=True, need_last_state=return_last_state)
        ctx.delta_softplus = delta_softplus
        ctx.has_z = z is not None
        ctx.chunk = r["chunk"]
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, r["out"], r["ckpt"])
        res = r["out_z"] if ctx.has_z else r["out"]
        if return_last_state:
            ctx.mark_non_differentiable(r["last_state"])
            return res, r["last_state"]
        return res

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, z, delta_bias, out, ckpt = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        A32 = A.float().contiguous()
        D32 = D.float() if D is not None else None
        db32 = delta_bias.float().contiguous() if delta_bias is not None else None
        g = ops_raw.scan_bwd(L.get_lib(), u, delta, A32, B, C, D32, z, db32, dout, out, ckpt, ctx.delta_softplus,
                             channel_last=False, chunk=ctx.chunk)
        return (g["du"], g["ddelta"], g["dA"].to(A.dtype), g["dB"].to(B.dtype), g["dC"].to(C.dtype),
                g["dD"].to(D.dtype) if D is not None else None,
                g["dz"] if z is not None else None,
                g["ddelta_bias"].to(delta_bias.dtype) if delta_bias is not None else None,
                None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state); last_state has shape (batch, dim, dstate)
    and its gradient is not considered in the backward pass (reference :76-83).

    The kernels keep at most 16 states per lane (SegMamba's d_state); wider state spaces (the reference accepts up to
    256, selective_scan.cpp:247) run as a sum over 16-state blocks - the scan is linear in (B, C) blocks - with the skip
    term in the first block and the gate applied to the sum."""
    dstate = A.shape[-1]
    if B.dim() == 2 or C.dim() == 2:
        return _scan_constant_bc(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
    if dstate <= 16:
        return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
    if dstate > 256:
        raise RuntimeError("selective_scan only supports state dimension <= 256")        # reference selective_scan.cpp:247
    y, lasts = None, []
    for n0 in range(0, dstate, 16):
        blk = slice(n0, min(n0 + 16, dstate))
        r = SelectiveScanFn.apply(u, delta, A[:, blk], B[..., blk, :], C[..., blk, :], D if n0 == 0 else None, None,
                                  delta_bias, delta_softplus, return_last_state)
        if return_last_state:
            r, last = r
            lasts.append(last)
        y = r.float() if y is None else y + r.float()
    if z is not None:
        y = y * torch.nn.functional.silu(z.float())
    y = y.to(u.dtype)
    return (y, torch.cat(lasts, dim=-1)) if return_last_state else y


def _scan_constant_bc(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state):
    """Constant (per-channel) B and / or C - (dim, dstate) instead of input-dependent (batch, [groups,] dstate, seqlen); reference
    selective_scan_interface.py:104-110,122-133 and selective_scan_fwd_kernel.cuh:223-233.  Not on the SegMamba path, so no kernel
    variant: the scan is a sum over states of single-state scans, a constant B[d, n] scales that state's input channel-wise
    (u B[d, n] against an all-ones B_t), a constant C[d, n] scales its output; the skip term and the gate are applied to the
    sum.  Everything around the kernel calls is ordinary autograd, so B / C receive their (dim, dstate) gradients."""
    if A.is_complex():
        raise RuntimeError("complex A is not supported (not on the SegMamba path)")
    batch, dim, L = u.shape
    dstate = A.shape[-1]
    ones = u.new_ones(batch, 1, L)
    y, lasts = None, []
    for n in range(dstate):
        un = u if B.dim() > 2 else (u.float() * B[:, n].float()[None, :, None]).to(u.dtype)
        Bn = B[..., n:n + 1, :] if B.dim() > 2 else ones
        Cn = C[..., n:n + 1, :] if C.dim() > 2 else ones
        r = SelectiveScanFn.apply(un, delta, A[:, n:n + 1], Bn, Cn, None, None, delta_bias, delta_softplus, return_last_state)
        if return_last_state:
            r, last = r
            lasts.append(last)
        r = r.float()
        if C.dim() == 2:
            r = r * C[:, n].float()[None, :, None]
        y = r if y is None else y + r
    if D is not None:
        y = y + u.float() * D.float()[None, :, None]
    if z is not None:
        y = y * torch.nn.functional.silu(z.float())
    y = y.to(u.dtype)
    return (y, torch.cat(lasts, dim=-1)) if return_last_state else y


_RECOMPUTE = os.environ.get("SEGM_RECOMPUTE", "0") == "1"     # reference trade: recompute conv output / delta in backward
_ADD3 = os.environ.get("SEGM_ADD3", "1") == "1"               # three-way sums of the v3 block as one pass (segm_add3)
_FUSED_CONV1D = os.environ.get("SEGM_SCAN_FUSED_CONV1D", "0") == "1"     # conv1d + SiLU inside the scan passes (opt-in, slower)
_FUSED_DTPROJ = os.environ.get("SEGM_SCAN_FUSED_DTPROJ", "0") == "1"     # dt_proj inside the scan passes (opt-in)


# ---------------------------------------------------------------------------------------------------------
# fused inner pipeline: conv1d+SiLU -> x_proj -> dt_proj -> selective scan (gated by silu(z))
# ---------------------------------------------------------------------------------------------------------
class MambaInnerCore(torch.autograd.Function):
    """out_z = scan(silu(conv1d(x)), softplus(dt_proj(x_proj(.)) + bias), A, B(.), C(.), D) * silu(z).

    xz is (batch, 2*dim, seqlen) when `channel_last` is False (the reference layout) and
    (batch, seqlen, 2*dim) when True; the result has the layout of one half of xz.  `time_order` /
    `nslices` select the logical direction (lib.TIME_*).  The reference recomputes the conv output and delta in backward
    (checkpoint_lvl=1, :161,218-219,238-241) to save memory on 16 - 80 GB devices; an MI355X has 288 GB, so by default they
    are KEPT (2 x 100 MB per direction at stage 0 of a 2 x 128^3 step, 1.6 GB for the whole network) and the backward starts
    without the conv1d / dt_proj launches.  SEGM_RECOMPUTE=1 restores the reference's trade.
    """

    @staticmethod
    @_custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                B_proj_bias, C_proj_bias, delta_softplus, channel_last, time_order, nslices, train=True):
        lib = L.get_lib()
        if torch.is_autocast_enabled():
            act_dtype = torch.get_autocast_dtype("cuda")
            from .param_bank import low_precision
            x_proj_weight = low_precision(x_proj_weight, act_dtype)
            delta_proj_weight = low_precision(delta_proj_weight, act_dtype)
        cdim = 2 if channel_last else 1
        if (xz.stride(1) if channel_last else xz.stride(2)) != 1 and xz.stride(cdim) != 1:
            xz = xz.contiguous()
        dim = xz.shape[cdim] // 2
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        if A.is_complex():
            raise RuntimeError("complex A is not supported (not on the SegMamba path)")
        w32 = conv1d_weight.reshape(dim, -1).float().contiguous()
        cb32 = conv1d_bias.float().contiguous() if conv1d_bias is not None else None
        A32 = A.float().contiguous()
        D32 = D.float().contiguous() if D is not None else None
        db32 = delta_bias.float().contiguous() if delta_bias is not None else None
        x, z = xz.split(dim, dim=cdim)
        conv_out = ops_raw.conv1d_fwd(lib, x, w32, cb32, True, channel_last=channel_last, time_order=time_order,
                                      nslices=nslices)
        if _rows_route(conv_out, channel_last) and B_proj_bias is None and C_proj_bias is None:
            x_dbl, delta, Bv, Cv = _project_rows(conv_out, x_proj_weight, delta_proj_weight, R, N)
        else:
            x_dbl, delta, Bv, Cv = _project(conv_out, x_proj_weight, delta_proj_weight, R, N, channel_last,
                                            B_proj_bias, C_proj_bias)
        # the un-gated y and the state checkpoints are only what the backward starts from: inference skips both stores.
        # `train` is decided by the caller (_inner): ctx.needs_input_grad stays True for parameters under no_grad().
        r = ops_raw.scan_fwd(lib, conv_out, delta, A32, Bv, Cv, D32, z, db32, delta_softplus,
                             channel_last=channel_last, time_order=time_order, nslices=nslices,
                             need_out=train, need_ckpt=train)
        keep = train and not _RECOMPUTE
        ctx.cfg = (bool(delta_softplus), bool(channel_last), int(time_order), int(nslices), r["chunk"], R, N,
                   B_proj_bias is not None, C_proj_bias is not None, keep)
        ctx.save_for_backward(xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, A, D,
                              delta_bias, B_proj_bias, C_proj_bias, r["out"], r["ckpt"],
                              conv_out if keep else None, delta if keep else None)
        return r["out_z"]

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = L.get_lib()
        (xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, A, D, delta_bias,
         B_proj_bias, C_proj_bias, out, ckpt, conv_out, delta) = ctx.saved_tensors
        delta_softplus, channel_last, time_order, nslices, chunk, R, N, has_Bb, has_Cb, keep = ctx.cfg
        cdim = 2 if channel_last else 1
        dim = xz.shape[cdim] // 2
        batch = xz.shape[0]
        seqlen = xz.shape[1] if channel_last else xz.shape[2]
        w32 = conv1d_weight.reshape(dim, -1).float().contiguous()
        cb32 = conv1d_bias.float().contiguous() if conv1d_bias is not None else None
        A32 = A.float().contiguous()
        D32 = D.float().contiguous() if D is not None else None
        db32 = delta_bias.float().contiguous() if delta_bias is not None else None
        x, z = xz.split(dim, dim=cdim)
        if (dout.stride(1) if channel_last else dout.stride(2)) != 1 and dout.stride(cdim) != 1:
            dout = dout.contiguous()
        rows_route = x_dbl.shape[1] != R + 2 * N           # the forward kept x_dbl padded: library projection route
        if keep:                                           # conv output and delta kept by the forward: B / C are views of x_dbl
            Bv, Cv = _bc_views(x_dbl, batch, seqlen, R, N, channel_last, B_proj_bias, C_proj_bias)
        else:                                              # recompute them (checkpoint_lvl 1)
            conv_out = ops_raw.conv1d_fwd(lib, x, w32, cb32, True, channel_last=channel_last, time_order=time_order,
                                          nslices=nslices)
            if rows_route:
                _, delta, Bv, Cv = _project_rows(conv_out, x_proj_weight, delta_proj_weight, R, N, x_dbl=x_dbl)
            else:
                _, delta, Bv, Cv = _project(conv_out, x_proj_weight, delta_proj_weight, R, N, channel_last,
                                            B_proj_bias, C_proj_bias, x_dbl=x_dbl)
        dxz = torch.empty_like(xz, memory_format=torch.contiguous_format)
        dx, dz = dxz.split(dim, dim=cdim)                 # dx / dz written in place (reference :244-245)
        # dB / dC: offered the columns of the x_proj gradient operand as destinations (taken by the deterministic kernel)
        dx_dbl, dBv, dCv = _dx_dbl_targets(x_dbl, batch, seqlen, R, N, channel_last)
        g = ops_raw.scan_bwd(lib, conv_out, delta, A32, Bv, Cv, D32, z, db32, dout, out, ckpt, delta_softplus,
                             channel_last=channel_last, time_order=time_order, nslices=nslices, chunk=chunk, dz=dz,
                             dB=dBv, dC=dCv)
        dconv_out, ddelta = g["du"], g["ddelta"]
        # (b*l, .) matrices for the projection gradients
        if channel_last:
            ddelta2 = ddelta.reshape(batch * seqlen, dim)                       # (bl, d)
            conv2 = conv_out.reshape(batch * seqlen, dim)
            dconv2 = dconv_out.reshape(batch * seqlen, dim)
        else:
            ddelta2 = ddelta.permute(0, 2, 1).reshape(batch * seqlen, dim)
            conv2 = conv_out.permute(0, 2, 1).reshape(batch * seqlen, dim)
            dconv2 = dconv_out.permute(0, 2, 1).reshape(batch * seqlen, dim)
        _dx_dbl_finish(dx_dbl, g, batch, seqlen, R, N, channel_last)
        c0 = _bc_col(x_dbl, R, N)
        dB_proj_bias = dx_dbl[:, c0:c0 + N].float().sum(0).to(B_proj_bias.dtype) if has_Bb else None
        dC_proj_bias = dx_dbl[:, c0 + N:c0 + 2 * N].float().sum(0).to(C_proj_bias.dtype) if has_Cb else None
        ddelta_proj_weight = tn_matmul(ddelta2, x_dbl[:, :R])                   # (d, R)    reference :272
        if rows_route:
            R4, P4, P8 = _rows_cols(R, N)
            # (bl, R) = ddelta2 @ dt_proj_weight (reference :273), with the zero padding columns up to R4
            ops_raw.linear_rows(lib, ddelta2, _pk(delta_proj_weight, ("dt_proj_t_rows", R4), lambda t: _pad_rows(t.t(), R4)), out=dx_dbl[:, :R4])
            dx_proj_weight = _x_proj_grad_rows(tn_matmul(dx_dbl[:, :P4], conv2), R, N)          # (R+2N, d) reference :275
            wx_t = _pk(x_proj_weight, ("x_proj_t4", P8), lambda t: _x_proj_rows4(t, R, N).t().contiguous())     # (d, P8)
            dconv2 = ops_raw.linear_rows(lib, dx_dbl, wx_t, out=dconv2, accumulate=True)        # reference :276
        else:
            dx_dbl[:, :R] = ddelta2 @ delta_proj_weight                         # (bl, R)   reference :273
            dx_proj_weight = tn_matmul(dx_dbl, conv2)                          # (R+2N, d) reference :275
            dconv2 = torch.addmm(dconv2, dx_dbl, x_proj_weight)                 # (bl, d)   reference :276
        dconv_full = dconv2.reshape(batch, seqlen, dim)
        if not channel_last:
            dconv_full = dconv_full.permute(0, 2, 1)                             # strided (b, d, l) view
        _, dconv_w, dconv_b = ops_raw.conv1d_bwd(lib, x, w32, cb32, dconv_full, True, channel_last=channel_last,
                                                 time_order=time_order, nslices=nslices, dx=dx)
        dconv_w = dconv_w.reshape(conv1d_weight.shape).to(conv1d_weight.dtype)
        dconv_b = dconv_b.to(conv1d_bias.dtype) if conv1d_bias is not None else None
        return (dxz, dconv_w, dconv_b, dx_proj_weight.to(x_proj_weight.dtype),
                ddelta_proj_weight.to(delta_proj_weight.dtype), g["dA"].to(A.dtype),
                g["dD"].to(D.dtype) if D is not None else None,
                g["ddelta_bias"].to(delta_bias.dtype) if delta_bias is not None else None,
                dB_proj_bias, dC_proj_bias, None, None, None, None, None)


class MambaInnerCore3(torch.autograd.Function):
    """The three directions of a `Mamba(bimamba_type="v3")` layer as ONE autograd node (reference mamba_simple.py:216-264 calls
    `mamba_inner_fn_no_out_proj` three times: as stored, time-reversed, slice-interleaved - three parameter sets on one `xz`).

    Same arithmetic per direction as `MambaInnerCore` on channel-last tensors; what the node buys is launch structure: the three
    selective scans of the forward (and of the backward) are ONE grid with a direction axis
    (`segm_selective_scan_{fwd,bwd}_multi`) - 3 x the waves where one direction cannot fill the 1024 SIMDs (stages 1 - 3:
    12.6 M, 3.1 M, 0.8 M channel-steps) and a third of the scan launches - and the three `dxz` contributions are summed inside
    the node.  Arguments: xz (B, L, 2D), nslices, train, then for each direction (conv1d_weight, conv1d_bias, x_proj_weight,
    delta_proj_weight, A, D, delta_bias)."""

    ORDERS = (L.TIME_FORWARD, L.TIME_REVERSED, L.TIME_INTERLEAVED)

    @staticmethod
    @_custom_fwd
    def forward(ctx, xz, nslices, train, *params):
        lib = L.get_lib()
        assert len(params) == 21
        sets = [params[7 * i:7 * i + 7] for i in range(3)]
        # 16-bit activations with fp32 master weights (autocast, or a caller that feeds 16-bit tokens): the projections take the
        # step's 16-bit copies of the masters (param_bank.low_precision), their gradients go back in fp32
        act_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else (
            xz.dtype if xz.dtype in (torch.bfloat16, torch.float16) else None)
        if xz.stride(2) != 1:
            xz = xz.contiguous()
        batch, seqlen, dim2 = xz.shape
        dim = dim2 // 2
        x, z = xz.split(dim, dim=2)
        keep = train and not _RECOMPUTE
        calls, per_dir = [], []
        conv_outs = ops_raw.conv1d_fwd_multi(lib, [
            dict(x=x, weight=st[0].reshape(dim, -1).float().contiguous(), bias=st[1].float().contiguous() if st[1] is not None else None,
                 silu=True, channel_last=True, time_order=MambaInnerCore3.ORDERS[i],
                 nslices=nslices if MambaInnerCore3.ORDERS[i] == L.TIME_INTERLEAVED else 1) for i, st in enumerate(sets)])
        for i, (conv_w, conv_b, xw, dtw, A, D, dbias) in enumerate(sets):
            if act_dtype is not None and xw.dtype != act_dtype:
                from .param_bank import low_precision
                xw, dtw = low_precision(xw, act_dtype), low_precision(dtw, act_dtype)
            R, N = dtw.shape[1], A.shape[-1]
            ns = nslices if MambaInnerCore3.ORDERS[i] == L.TIME_INTERLEAVED else 1
            conv_out = conv_outs[i]
            # dt_proj inside the scan launches (opt-in): the stored-delta forms only (with recompute the backward would form delta
            # through the other kernel and the two roundings could differ by an ulp of the 16-bit type)
            dt_in_scan = _FUSED_DTPROJ and not _FUSED_CONV1D and R <= 8 and _rows_route(conv_out, True) and (keep or not train) and \
                ops_raw.scan_fused_conv_supported(lib, batch, dim, seqlen, ns, MambaInnerCore3.ORDERS[i])
            if _rows_route(conv_out, True):
                x_dbl, delta, Bv, Cv = _project_rows(conv_out, xw, dtw, R, N, delta_in_scan=dt_in_scan)
            else:
                x_dbl, delta, Bv, Cv = _project(conv_out, xw, dtw, R, N, True, None, None)
            calls.append(dict(u=conv_out, delta=delta, A=A.float().contiguous(), B=Bv, C=Cv,
                              D=D.float().contiguous() if D is not None else None, z=z,
                              delta_bias=dbias.float().contiguous() if dbias is not None else None, delta_softplus=True,
                              channel_last=True, time_order=MambaInnerCore3.ORDERS[i], nslices=ns, need_out=train, need_ckpt=train))
            if dt_in_scan:
                calls[-1].update(dt_x=x_dbl.view(batch, seqlen, -1)[:, :, :R],
                                 dt_weight=_pk(dtw, ("dt_proj_f32", R), lambda t: t.float().contiguous()))
            if _FUSED_CONV1D and ops_raw.scan_fused_conv_supported(lib, batch, dim, seqlen, ns, calls[-1]["time_order"]):
                # the north star's "conv1d fused into the scan launch": the passes read x and form u themselves (bit-identical).
                # conv_out is still produced above - x_proj needs every channel of it before the scan can start - so this only
                # moves work into the instruction-bound scan passes; measured slower, hence opt-in
                calls[-1].update(u=x, conv_weight=conv_w.reshape(dim, -1), conv_bias=conv_b)
            per_dir.append((x_dbl, xw, dtw, conv_out, delta, R, N, ns))
        rs = ops_raw.scan_fwd_multi(lib, calls)
        saved = [xz]
        for (conv_w, conv_b, xw0, dtw0, A, D, dbias), (x_dbl, xw, dtw, conv_out, delta, R, N, ns), r in zip(sets, per_dir, rs):
            saved += [conv_w, conv_b, x_dbl, xw, dtw, A, D, dbias, r["out"], r["ckpt"],
                      conv_out if keep else None, delta if keep else None]
        ctx.cfg = (int(nslices), [r["chunk"] for r in rs], [(p[5], p[6]) for p in per_dir], keep,
                   [(st[2].dtype, st[3].dtype) for st in sets])         # the masters' dtypes: weight gradients go back in them
        ctx.save_for_backward(*saved)
        return tuple(r["out_z"] for r in rs)

    @staticmethod
    @_custom_bwd
    def backward(ctx, *douts):
        lib = L.get_lib()
        nslices, chunks, rn, keep, wdt = ctx.cfg
        saved = ctx.saved_tensors
        xz = saved[0]
        batch, seqlen, dim2 = xz.shape
        dim = dim2 // 2
        x, z = xz.split(dim, dim=2)
        dirs, calls = [], []
        for i in range(3):
            (conv_w, conv_b, x_dbl, xw, dtw, A, D, dbias, out, ckpt, conv_out, delta) = saved[1 + 12 * i:13 + 12 * i]
            R, N = rn[i]
            order = MambaInnerCore3.ORDERS[i]
            ns = nslices if order == L.TIME_INTERLEAVED else 1
            w32 = conv_w.reshape(dim, -1).float().contiguous()
            cb32 = conv_b.float().contiguous() if conv_b is not None else None
            rows_route = x_dbl.shape[1] != R + 2 * N
            if keep:
                Bv, Cv = _bc_views(x_dbl, batch, seqlen, R, N, True, None, None)
            else:
                conv_out = ops_raw.conv1d_fwd(lib, x, w32, cb32, True, channel_last=True, time_order=order, nslices=ns)
                if rows_route:
                    _, delta, Bv, Cv = _project_rows(conv_out, xw, dtw, R, N, x_dbl=x_dbl)
                else:
                    _, delta, Bv, Cv = _project(conv_out, xw, dtw, R, N, True, None, None, x_dbl=x_dbl)
            dout = douts[i]
            if dout.stride(2) != 1:
                dout = dout.contiguous()
            dxz = torch.empty_like(xz, memory_format=torch.contiguous_format)
            dx, dz = dxz.split(dim, dim=2)
            dx_dbl, dBv, dCv = _dx_dbl_targets(x_dbl, batch, seqlen, R, N, True)
            calls.append(dict(u=conv_out, delta=delta, A=A.float().contiguous(), B=Bv, C=Cv,
                              D=D.float().contiguous() if D is not None else None, z=z,
                              delta_bias=dbias.float().contiguous() if dbias is not None else None, dout=dout, out=out, ckpt=ckpt,
                              delta_softplus=True, channel_last=True, time_order=order, nslices=ns, chunk=chunks[i], dz=dz,
                              dB=dBv, dC=dCv))
            dirs.append((conv_w, conv_b, x_dbl, xw, dtw, A, D, dbias, conv_out, R, N, order, ns, w32, cb32, rows_route, dxz, dx, dx_dbl))
        gs = ops_raw.scan_bwd_multi(lib, calls)
        grads, dxz_sum, ccalls, part = [], None, [], []
        for i, ((conv_w, conv_b, x_dbl, xw, dtw, A, D, dbias, conv_out, R, N, order, ns, w32, cb32, rows_route, dxz, dx, dx_dbl), g) in enumerate(zip(dirs, gs)):
            dconv2 = g["du"].reshape(batch * seqlen, dim)
            ddelta2 = g["ddelta"].reshape(batch * seqlen, dim)
            conv2 = conv_out.reshape(batch * seqlen, dim)
            _dx_dbl_finish(dx_dbl, g, batch, seqlen, R, N, True)
            ddelta_proj_weight = tn_matmul(ddelta2, x_dbl[:, :R])
            if rows_route:
                R4, P4, P8 = _rows_cols(R, N)
                ops_raw.linear_rows(lib, ddelta2, _pk(dtw, ("dt_proj_t_rows", R4), lambda t: _pad_rows(t.t(), R4)), out=dx_dbl[:, :R4])
                dx_proj_weight = _x_proj_grad_rows(tn_matmul(dx_dbl[:, :P4], conv2), R, N)
                wx_t = _pk(xw, ("x_proj_t4", P8), lambda t: _x_proj_rows4(t, R, N).t().contiguous())
                dconv2 = ops_raw.linear_rows(lib, dx_dbl, wx_t, out=dconv2, accumulate=True)
            else:
                dx_dbl[:, :R] = ddelta2 @ dtw
                dx_proj_weight = tn_matmul(dx_dbl, conv2)
                dconv2 = torch.addmm(dconv2, dx_dbl, xw)
            ccalls.append(dict(x=x, weight=w32, bias=cb32, dout=dconv2.reshape(batch, seqlen, dim), silu=True, channel_last=True,
                               time_order=order, nslices=ns, dx=dx))
            part.append((dx_proj_weight, ddelta_proj_weight))
        cres = ops_raw.conv1d_bwd_multi(lib, ccalls)
        # the three dxz contributions: one pass a + b + c (segm_add3) instead of two in-place binary adds (SEGM_ADD3=0: those)
        all_dxz = [d[16] for d in dirs]
        use3 = _ADD3 and len(all_dxz) == 3 and ops_raw.add3_supported(*all_dxz)
        if use3:
            dxz_sum = ops_raw.add3(lib, *all_dxz, out=all_dxz[0])
        for i, ((conv_w, conv_b, x_dbl, xw, dtw, A, D, dbias, conv_out, R, N, order, ns, w32, cb32, rows_route, dxz, dx, dx_dbl), g) in enumerate(zip(dirs, gs)):
            _, dconv_w, dconv_b = cres[i]
            dx_proj_weight, ddelta_proj_weight = part[i]
            if not use3:
                dxz_sum = dxz if dxz_sum is None else dxz_sum.add_(dxz)
            grads += [dconv_w.reshape(conv_w.shape).to(conv_w.dtype), dconv_b.to(conv_b.dtype) if conv_b is not None else None,
                      dx_proj_weight.to(wdt[i][0]), ddelta_proj_weight.to(wdt[i][1]), g["dA"].to(A.dtype),
                      g["dD"].to(D.dtype) if D is not None else None,
                      g["ddelta_bias"].to(dbias.dtype) if dbias is not None else None]
        return (dxz_sum, None, None, *grads)


def _bc_col(x_dbl, R, N):
    """first B column of an x_dbl / dx_dbl matrix: R in the reference layout (dt | B | C), R rounded up to 4 in the padded
    layout of the row-streaming projection route (dt | 0.. | B | C | 0..; see _project_rows)"""
    return R if x_dbl.shape[1] == R + 2 * N else -(-R // 4) * 4


def _bc_views(x_dbl, batch, seqlen, R, N, channel_last, B_proj_bias, C_proj_bias):
    """B_t / C_t as the scan expects them: strided views of x_dbl (b*l, R + 2N), or of its padded form"""
    v3 = x_dbl.view(batch, seqlen, x_dbl.shape[1])
    c0 = _bc_col(x_dbl, R, N)
    Bv, Cv = v3[:, :, c0:c0 + N], v3[:, :, c0 + N:c0 + 2 * N]
    if not channel_last:
        Bv, Cv = Bv.permute(0, 2, 1), Cv.permute(0, 2, 1)
    if B_proj_bias is not None:
        Bv = Bv + (B_proj_bias.to(Bv.dtype) if channel_last else B_proj_bias.to(Bv.dtype)[:, None])
    if C_proj_bias is not None:
        Cv = Cv + (C_proj_bias.to(Cv.dtype) if channel_last else C_proj_bias.to(Cv.dtype)[:, None])
    return Bv, Cv


def _rows_route(conv_out, channel_last) -> bool:
    """The library's row-streaming projection kernel (csrc/linear.hip) for x_proj / dt_proj and their data gradients: opt-in
    (SEGM_LINEAR_HIP=1, see linear.py), channel-last 16-bit activations with enough rows only."""
    from . import linear as LN
    return bool(LN._ROWS_HIP and channel_last and LN._on_device(conv_out) and conv_out.dtype in (torch.bfloat16, torch.float16)
                and conv_out.shape[0] * conv_out.shape[1] >= LN._ROWS_MIN and conv_out.shape[2] % 8 == 0
                and conv_out.shape[2] <= 192 and conv_out.is_contiguous())


def _pk(w, tag, fn):
    from .param_bank import packed
    return packed(w, tag, fn)


def _pad_cols(w, cols):
    """w (r, c) -> (r, cols) with zero columns appended"""
    if w.shape[1] == cols:
        return w.contiguous()
    out = w.new_zeros(w.shape[0], cols)
    out[:, :w.shape[1]] = w
    return out


def _pad_rows(w, rows):
    """w (r, c) -> (rows, c) with zero rows appended"""
    if w.shape[0] == rows:
        return w.contiguous()
    out = w.new_zeros(rows, w.shape[1])
    out[:w.shape[0]] = w
    return out


def _rows_cols(R, N):
    """column layout of x_dbl / dx_dbl on the row-streaming route: dt in [0, R), zeros up to R4 = R rounded up to 4, B in
    [R4, R4 + N), C behind it, zeros up to P8 (a multiple of 8; 40 columns for R = 3 or 6 with N = 16, as many as the plain
    padding of R + 2N needs).  B starts on a 4-column group so that the kernel that writes the dt gradient (four columns per
    lane) never touches a B column: the scan backward stores dB / dC straight into these columns of dx_dbl."""
    R4 = -(-R // 4) * 4
    P4 = R4 + 2 * N
    return R4, P4, -(-P4 // 8) * 8


def _x_proj_rows4(w, R, N):
    """x_proj weight (R + 2N, d) -> (P8, d) in the padded column layout"""
    R4, P4, P8 = _rows_cols(R, N)
    out = w.new_zeros(P8, w.shape[1])
    out[:R] = w[:R]
    out[R4:P4] = w[R:]
    return out


def _project_rows(conv_out, x_proj_weight, delta_proj_weight, R, N, x_dbl=None, delta_in_scan=False):
    """`_project` for channel-last activations through segm_linear_rows.  x_dbl is kept in the padded column layout of
    `_rows_cols` (the extra columns are zero: zero weight rows) so that it can be both an output and - its first 8-column
    group, against a zero-padded dt_proj weight - an input of the kernel.  `delta_in_scan`: delta comes back as an empty tensor
    the scan's apply pass fills (ops_raw.scan_fwd: dt_x = x_dbl's first R columns)."""
    lib = L.get_lib()
    batch, seqlen, dim = conv_out.shape
    R4, P4, P8 = _rows_cols(R, N)
    R8 = -(-R // 8) * 8
    if x_dbl is None:
        x_dbl = ops_raw.linear_rows(lib, conv_out.reshape(batch * seqlen, dim),
                                    _pk(x_proj_weight, ("x_proj_rows4", P8), lambda t: _x_proj_rows4(t, R, N)))
    if delta_in_scan:
        delta = torch.empty(batch, seqlen, dim, dtype=conv_out.dtype, device=conv_out.device)
    else:
        wdt = _pk(delta_proj_weight, ("dt_proj_cols", R8), lambda t: _pad_cols(t, R8))
        delta = ops_raw.linear_rows(lib, x_dbl[:, :R8], wdt).reshape(batch, seqlen, dim)
    v3 = x_dbl.view(batch, seqlen, P8)
    return x_dbl, delta, v3[:, :, R4:R4 + N], v3[:, :, R4 + N:P4]


def _dx_dbl_targets(x_dbl, batch, seqlen, R, N, channel_last):
    """the x_proj gradient operand dx_dbl (same shape as x_dbl) and its dB / dC column windows in the layout the scan backward
    expects - offered to it as destinations of the activations' own type (ops_raw.scan_bwd: `dbc_native`)"""
    dx_dbl = torch.empty_like(x_dbl)
    c0 = _bc_col(x_dbl, R, N)
    v3 = dx_dbl.view(batch, seqlen, dx_dbl.shape[1])
    dBv, dCv = v3[:, :, c0:c0 + N], v3[:, :, c0 + N:c0 + 2 * N]
    if not channel_last:
        dBv, dCv = dBv.permute(0, 2, 1), dCv.permute(0, 2, 1)
    return dx_dbl, dBv, dCv


def _dx_dbl_finish(dx_dbl, g, batch, seqlen, R, N, channel_last):
    """dB / dC into their columns unless the scan already wrote them there; zero padding columns"""
    c0 = _bc_col(dx_dbl, R, N)
    if not g["dbc_native"]:
        if channel_last:
            dB2, dC2 = g["dB"].reshape(batch * seqlen, N), g["dC"].reshape(batch * seqlen, N)
        else:
            dB2 = g["dB"].reshape(batch, N, seqlen).permute(0, 2, 1).reshape(batch * seqlen, N)
            dC2 = g["dC"].reshape(batch, N, seqlen).permute(0, 2, 1).reshape(batch * seqlen, N)
        dx_dbl[:, c0:c0 + N] = dB2
        dx_dbl[:, c0 + N:c0 + 2 * N] = dC2
    if dx_dbl.shape[1] > c0 + 2 * N:
        dx_dbl[:, c0 + 2 * N:] = 0


def _x_proj_grad_rows(dw, R, N):
    """weight gradient in the padded column layout (P4, d) -> (R + 2N, d)"""
    R4 = -(-R // 4) * 4
    return dw if R4 == R else torch.cat([dw[:R], dw[R4:R4 + 2 * N]], 0)


def _project(conv_out, x_proj_weight, delta_proj_weight, R, N, channel_last, B_proj_bias, C_proj_bias, x_dbl=None):
    """x_dbl = x_proj(conv_out) as (b*l, R+2N); delta = dt_proj(x_dbl[:, :R]) in conv_out's layout;
    B / C as strided views of x_dbl in the layout the scan expects (no transposing copy)."""
    if channel_last:
        batch, seqlen, dim = conv_out.shape
        if x_dbl is None:
            x_dbl = F.linear(conv_out.reshape(batch * seqlen, dim), x_proj_weight)
        delta = F.linear(x_dbl[:, :R], delta_proj_weight).reshape(batch, seqlen, dim)
        v3 = x_dbl.view(batch, seqlen, R + 2 * N)
        Bv, Cv = v3[:, :, R:R + N], v3[:, :, R + N:]
    else:
        batch, dim, seqlen = conv_out.shape
        if x_dbl is None:
            x_dbl = F.linear(conv_out.permute(0, 2, 1).reshape(batch * seqlen, dim), x_proj_weight)
        delta = (delta_proj_weight @ x_dbl[:, :R].t()).reshape(dim, batch, seqlen).permute(1, 0, 2).contiguous()
        v3 = x_dbl.view(batch, seqlen, R + 2 * N)
        Bv, Cv = v3[:, :, R:R + N].permute(0, 2, 1), v3[:, :, R + N:].permute(0, 2, 1)
    if B_proj_bias is not None:
        Bv = Bv + (B_proj_bias.to(Bv.dtype) if channel_last else B_proj_bias.to(Bv.dtype)[:, None])
    if C_proj_bias is not None:
        Cv = Cv + (C_proj_bias.to(Cv.dtype) if channel_last else C_proj_bias.to(Cv.dtype)[:, None])
    return x_dbl, delta, Bv, Cv


def _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
           B_proj_bias, C_proj_bias, delta_softplus, channel_last=False, time_order=L.TIME_FORWARD, nslices=1):
    if B is not None or C is not None:
        return _inner_constant_bc(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                                  B_proj_bias, C_proj_bias, delta_softplus, channel_last, time_order, nslices)
    # inference (no_grad, or nothing requires a gradient): the forward skips the stores only a backward would read.
    # is_grad_enabled() is always False inside Function.forward, so the decision is taken here.
    train = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D,
                                                    delta_bias, B_proj_bias, C_proj_bias))
    return MambaInnerCore.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                                B_proj_bias, C_proj_bias, delta_softplus, channel_last, time_order, nslices, train)


def _inner_constant_bc(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                       B_proj_bias, C_proj_bias, delta_softplus, channel_last, time_order, nslices):
    """The inner pipeline with a constant (dim, dstate) B and / or C (reference :170-213 with `is_variable_B/C` False): x_proj
    then has dt_rank + dstate * (number of input-dependent matrices) rows.  Off the SegMamba path: composed from the public ops
    (causal_conv1d_fn, the projections, selective_scan_fn), differentiated by autograd, reference layout only."""
    if channel_last or time_order != L.TIME_FORWARD:
        raise RuntimeError("constant B / C is implemented for the reference (batch, dim, seqlen) layout in forward time order")
    from .causal_conv1d_interface import causal_conv1d_fn
    dim = xz.shape[1] // 2
    batch, seqlen = xz.shape[0], xz.shape[2]
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz.split(dim, dim=1)
    conv_out = causal_conv1d_fn(x, conv1d_weight.reshape(dim, -1), conv1d_bias, "silu")
    x_dbl = F.linear(conv_out.permute(0, 2, 1).reshape(batch * seqlen, dim), x_proj_weight.to(conv_out.dtype))
    delta = (delta_proj_weight.to(x_dbl.dtype) @ x_dbl[:, :R].t()).reshape(dim, batch, seqlen).permute(1, 0, 2).contiguous()
    if B is None:
        B = x_dbl[:, R:R + N]
        if B_proj_bias is not None:
            B = B + B_proj_bias.to(B.dtype)
        B = B.reshape(batch, seqlen, N).permute(0, 2, 1).contiguous()
    if C is None:
        C = x_dbl[:, -N:]
        if C_proj_bias is not None:
            C = C + C_proj_bias.to(C.dtype)
        C = C.reshape(batch, seqlen, N).permute(0, 2, 1).contiguous()
    return selective_scan_fn(conv_out, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                               A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                               C_proj_bias=None, delta_softplus=True):
    """reference :627-633.  xz: (batch, 2*dim, seqlen) -> (batch, dim, seqlen)"""
    return _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                  B_proj_bias, C_proj_bias, delta_softplus)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                   out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                   C_proj_bias=None, delta_softplus=True):
    """reference :606-614 (MambaInnerFn :292-434): the inner pipeline followed by the output projection."""
    y = _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
               B_proj_bias, C_proj_bias, delta_softplus)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                     out_proj_weight, out_proj_bias,
                     A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                     C_proj_bias=None, delta_softplus=True):
    """reference :616-624 (BiMambaInnerFn :437-603): forward scan with A plus a scan of the reversed
    sequence with A_b sharing every other weight, summed, then the output projection."""
    y = _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
               B_proj_bias, C_proj_bias, delta_softplus)
    # the reference reverses u, delta, B, C, z *after* the (forward-order) conv and projections
    # (:478-486), i.e. only the recurrence runs right-to-left; expressed here by flipping its inputs/outputs
    y_b = _bimamba_reverse(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A_b, D, delta_bias,
                           B_proj_bias, C_proj_bias, delta_softplus)
    return F.linear((y + y_b).transpose(1, 2), out_proj_weight, out_proj_bias)


def _bimamba_reverse(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A_b, D, delta_bias,
                     B_proj_bias, C_proj_bias, delta_softplus):
    from .causal_conv1d_interface import causal_conv1d_fn
    dim = xz.shape[1] // 2
    R = delta_proj_weight.shape[1]
    N = A_b.shape[-1]
    x, z = xz.split(dim, dim=1)
    conv_out = causal_conv1d_fn(x, conv1d_weight.reshape(dim, -1), conv1d_bias, "silu")
    _, delta, Bv, Cv = _project(conv_out, x_proj_weight.to(conv_out.dtype), delta_proj_weight.to(conv_out.dtype),
                                R, N, False, B_proj_bias, C_proj_bias)
    y_b = selective_scan_fn(conv_out.flip(-1), delta.flip(-1), A_b, Bv.flip(-1), Cv.flip(-1), D, z.flip(-1),
                            delta_bias, delta_softplus)
    return y_b.flip(-1)
