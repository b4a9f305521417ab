"""Thin, allocation-owning wrappers around the C ABI: tensors in, tensors out, no autograd.

Every function takes the loaded library first so that the *same* marshalling code is exercised by
the product (`lib.get_lib()`, the HIP build) and by the CPU-emulation build the tests use to check
kernel logic without a GPU.  Layout is chosen per call:

  channel_last=True   u, delta, z, out ... are (B, L, D); B / C are (B, L, N) or (B, L, G, N)
  channel_last=False  the reference's layout: (B, D, L); B / C are (B, N, L) or (B, G, N, L)
                      (mamba/mamba_ssm/ops/selective_scan_interface.py:31-36)

Error behaviour mirrors the reference's TORCH_CHECKs (selective_scan.cpp:233-303,
causal_conv1d.cpp:136-170): shape / dtype / stride problems raise RuntimeError.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import torch

from . import lib as L


def _shape_bld(t: torch.Tensor, channel_last: bool):
    b, x, y = t.shape
    return (b, x, y) if channel_last else (b, y, x)        # -> (batch, seqlen, dim)


def _bc4(t: torch.Tensor, channel_last: bool) -> torch.Tensor:
    if t.dim() == 3:
        return t.unsqueeze(2) if channel_last else t.unsqueeze(1)
    if t.dim() != 4:
        raise RuntimeError("B / C must have 3 or 4 dimensions (input-dependent B and C only)")
    return t


def _check_seq(name, t, ref, dtype):
    if t is None:
        return
    if t.shape != ref.shape:
        raise RuntimeError(f"{name} must have shape {tuple(ref.shape)}, got {tuple(t.shape)}")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if t.device != ref.device:
        raise RuntimeError(f"{name} must be on {ref.device}")


def _check_vec(name, t, n, dev):
    if t is None:
        return
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    if t.numel() != n or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous with {n} elements")
    if t.device != dev:
        raise RuntimeError(f"{name} must be on {dev}")


def _fill_scan_args(a: L.ScanFwdArgs, u, delta, A, B, C, D, z, delta_bias, delta_softplus, channel_last,
                    time_order, nslices, chunk):
    batch, seqlen, dim = _shape_bld(u, channel_last)
    if A.dim() != 2 or A.shape[0] != dim:
        raise RuntimeError(f"A must be (dim={dim}, dstate)")
    if A.is_complex():
        raise RuntimeError("complex A is not supported (not on the SegMamba path)")
    dstate = A.shape[1]
    if dstate > 16:
        raise RuntimeError("selective_scan only supports state dimension <= 16 on this backend")
    B4, C4 = _bc4(B, channel_last), _bc4(C, channel_last)
    groups = B4.shape[2] if channel_last else B4.shape[1]
    exp_shape = (batch, seqlen, groups, dstate) if channel_last else (batch, groups, dstate, seqlen)
    for name, t in (("B", B4), ("C", C4)):
        if tuple(t.shape) != exp_shape:
            raise RuntimeError(f"{name} must have shape {exp_shape}, got {tuple(t.shape)}")
        if t.dtype != u.dtype:
            raise RuntimeError(f"{name} must have the dtype of u")
    if dim % groups != 0:
        raise RuntimeError("dim must be divisible by the number of B/C groups")
    _check_seq("delta", delta, u, u.dtype)
    _check_seq("z", z, u, u.dtype)
    _check_vec("A", A.reshape(-1), dim * dstate, u.device)
    _check_vec("D", D, dim, u.device)
    _check_vec("delta_bias", delta_bias, dim, u.device)
    if time_order == L.TIME_INTERLEAVED and (nslices <= 0 or seqlen % nslices != 0):
        raise RuntimeError(f"seqlen {seqlen} must be divisible by nslices {nslices}")
    a.batch, a.dim, a.dstate, a.n_groups, a.seqlen = batch, dim, dstate, groups, seqlen
    a.dtype = L.dtype_code(u)
    a.delta_softplus = int(bool(delta_softplus))
    a.time_order, a.nslices, a.chunk = int(time_order), int(nslices), int(chunk)
    a.u, a.delta, a.z = L.seq_view(u, channel_last), L.seq_view(delta, channel_last), L.seq_view(z, channel_last)
    a.B, a.C = L.bc_view(B4, channel_last), L.bc_view(C4, channel_last)
    a.A, a.D, a.delta_bias = A.data_ptr(), L.fptr(D), L.fptr(delta_bias)
    a.stream = L.stream_handle(u)
    return batch, seqlen, dim, dstate, groups, B4, C4


def scan_fwd(lib: L.SegmLib, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, *,
             channel_last=False, time_order=L.TIME_FORWARD, nslices=1, chunk=0, need_out=True,
             need_ckpt=False, need_last_state=False, ckpt_buf=None, conv_weight=None, conv_bias=None, dt_x=None, dt_weight=None):
    """-> dict(out, out_z, ckpt, last_state, chunk).  `out` is the un-gated y (None unless need_out or z is None).
    `dt_x` (batch, seqlen, R) rows of R <= 8 consecutive elements [a column window of x_dbl] + `dt_weight` (dim, R) fp32:
    delta = dt_weight . dt_x is formed inside the scan launches (reference selective_scan_interface.py:181-182 without its
    launch) and `delta` is an OUTPUT the apply pass fills, rounded to its dtype, for the backward (regular shapes, one group).
    `ckpt_buf`: a caller-owned fp32 buffer of segm_selective_scan_ckpt_bytes() for the checkpoints.
    `conv_weight` (dim, width) [+ `conv_bias` (dim)]: the causal depthwise conv1d + SiLU in front of the scan is computed inside the
    scan launches and `u` is its INPUT x (regular shapes with softplus and a gate only; results equal conv1d_fwd followed by the scan)."""
    a = L.ScanFwdArgs()
    r = _scan_fwd_prepare(lib, a, u, delta, A, B, C, D, z, delta_bias, delta_softplus, channel_last=channel_last,
                          time_order=time_order, nslices=nslices, chunk=chunk, need_out=need_out, need_ckpt=need_ckpt,
                          need_last_state=need_last_state, ckpt_buf=ckpt_buf, conv_weight=conv_weight, conv_bias=conv_bias,
                          dt_x=dt_x, dt_weight=dt_weight)
    lib.check(lib.dll.segm_selective_scan_fwd(a), "selective_scan_fwd")
    r.pop("_ws")
    return r


def scan_fused_conv_supported(lib: L.SegmLib, batch: int, dim: int, seqlen: int, nslices: int, time_order: int, chunk: int = 0,
                              dstate: int = 16) -> bool:
    """whether scan_fwd(..., conv_weight=) can run: the library's regular-shape condition"""
    return bool(lib.dll.segm_selective_scan_regular_shape(batch, dim, dstate, seqlen, chunk, int(time_order), int(nslices)))


def scan_fwd_multi(lib: L.SegmLib, calls):
    """Several forward scans of ONE geometry (the three directions of a Mamba v3 layer) as one C call: `calls` is a list of
    keyword dicts for `scan_fwd`; the library runs them as one grid with a direction axis when they are regular-shaped and share
    batch / dim / dstate / seqlen / dtype / chunk, else one after the other.  -> list of scan_fwd result dicts."""
    n = len(calls)
    arr = (L.ScanFwdArgs * n)()
    rs = [_scan_fwd_prepare(lib, arr[i], **c) for i, c in enumerate(calls)]
    lib.check(lib.dll.segm_selective_scan_fwd_multi(arr, n), "selective_scan_fwd_multi")
    for r in rs:
        r.pop("_ws")
    return rs


def _scan_fwd_prepare(lib, a, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, *,
                      channel_last=False, time_order=L.TIME_FORWARD, nslices=1, chunk=0, need_out=True,
                      need_ckpt=False, need_last_state=False, ckpt_buf=None, conv_weight=None, conv_bias=None,
                      dt_x=None, dt_weight=None):
    """fills the argument block `a` and allocates outputs / workspace; -> result dict (+ `_ws`, alive until the launch)"""
    batch, seqlen, dim, dstate, groups, B4, C4 = _fill_scan_args(
        a, u, delta, A, B, C, D, z, delta_bias, delta_softplus, channel_last, time_order, nslices, chunk)
    if chunk == 0:
        chunk = lib.dll.segm_selective_scan_default_chunk(batch, dim, seqlen)
        a.chunk = chunk
    dev = u.device
    out = torch.empty_like(u, memory_format=torch.contiguous_format) if (need_out or z is None) else None
    out_z = torch.empty_like(u, memory_format=torch.contiguous_format) if z is not None else None
    ckpt = None
    if need_ckpt:
        nbytes = lib.dll.segm_selective_scan_ckpt_bytes(batch, dim, dstate, seqlen)
        if ckpt_buf is not None:
            if ckpt_buf.dtype != torch.float32 or not ckpt_buf.is_contiguous() or ckpt_buf.numel() < nbytes // 4 or \
                    ckpt_buf.device != dev:
                raise RuntimeError("scan_fwd: ckpt_buf must be a contiguous fp32 tensor of segm_selective_scan_ckpt_bytes()")
            ckpt = ckpt_buf
        else:
            ckpt = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    last_state = torch.empty(batch, dim, dstate, dtype=torch.float32, device=dev) if need_last_state else None
    ws_bytes = lib.dll.segm_selective_scan_fwd_workspace_bytes(batch, dim, dstate, seqlen, chunk)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a.out, a.out_z = L.seq_view(out, channel_last), L.seq_view(out_z, channel_last)
    a.last_state, a.ckpt = L.fptr(last_state), L.fptr(ckpt)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    keep = [ws]
    if conv_weight is not None:
        if conv_weight.dim() != 2 or conv_weight.shape[0] != dim or not 2 <= conv_weight.shape[1] <= 4:
            raise RuntimeError("scan_fwd: conv_weight must be (dim, width) with width in [2, 4]")
        cw = conv_weight.detach().to(torch.float32).contiguous()
        cb = conv_bias.detach().to(torch.float32).contiguous() if conv_bias is not None else None
        a.conv_weight, a.conv_bias, a.conv_width = cw.data_ptr(), L.fptr(cb), cw.shape[1]
        keep += [cw, cb]
    if dt_x is not None:
        R = dt_x.shape[-1]
        if dt_x.dim() != 3 or tuple(dt_x.shape[:2]) != (batch, seqlen) or not 1 <= R <= 8 or dt_x.stride(2) != 1 or \
                dt_x.dtype != u.dtype or dt_x.device != dev:
            raise RuntimeError("scan_fwd: dt_x must be (batch, seqlen, R <= 8) of u's dtype with unit stride along R")
        if dt_weight is None or tuple(dt_weight.shape) != (dim, R) or dt_weight.dtype != torch.float32 or \
                not dt_weight.is_contiguous() or dt_weight.device != dev:
            raise RuntimeError("scan_fwd: dt_weight must be a contiguous fp32 (dim, R) tensor")
        a.dt_x, a.dt_stride_b, a.dt_stride_t = dt_x.data_ptr(), dt_x.stride(0), dt_x.stride(1)
        a.dt_weight, a.dt_rank = dt_weight.data_ptr(), R
        keep += [dt_x, dt_weight]
    return dict(out=out, out_z=out_z, ckpt=ckpt, last_state=last_state, chunk=chunk, _ws=keep)


def scan_bwd(lib: L.SegmLib, u, delta, A, B, C, D, z, delta_bias, dout, out, ckpt, delta_softplus, *,
             channel_last=False, time_order=L.TIME_FORWARD, nslices=1, chunk=0, du=None, ddelta=None, dz=None, dB=None, dC=None):
    """-> dict(du, ddelta, dA, dB, dC, dD, ddelta_bias, dz, dbc_native).  du / ddelta / dz may be pre-allocated views
    (e.g. halves of one dxz buffer, reference selective_scan_interface.py:244-245); dB / dC have the layout and rank of
    B / C and are fp32 (the reference's accumulation buffers) unless the caller offers destinations of u's own 16-bit type
    AND the launch takes the deterministic kernel (`dbc_native` in the result says which happened: see _scan_bwd_prepare)."""
    a = L.ScanBwdArgs()
    r = _scan_bwd_prepare(lib, a, u, delta, A, B, C, D, z, delta_bias, dout, out, ckpt, delta_softplus, channel_last=channel_last,
                          time_order=time_order, nslices=nslices, chunk=chunk, du=du, ddelta=ddelta, dz=dz, dB=dB, dC=dC)
    lib.check(lib.dll.segm_selective_scan_bwd(a), "selective_scan_bwd")
    r.pop("_ws")
    return r


def scan_bwd_multi(lib: L.SegmLib, calls):
    """the backward counterpart of scan_fwd_multi: `calls` = keyword dicts for `scan_bwd`"""
    n = len(calls)
    arr = (L.ScanBwdArgs * n)()
    rs = [_scan_bwd_prepare(lib, arr[i], **c) for i, c in enumerate(calls)]
    lib.check(lib.dll.segm_selective_scan_bwd_multi(arr, n), "selective_scan_bwd_multi")
    for r in rs:
        r.pop("_ws")
    return rs


def _scan_bwd_prepare(lib, a, u, delta, A, B, C, D, z, delta_bias, dout, out, ckpt, delta_softplus, *,
                      channel_last=False, time_order=L.TIME_FORWARD, nslices=1, chunk=0, du=None, ddelta=None, dz=None,
                      dB=None, dC=None):
    """`dB` / `dC`: optional caller-owned tensors shaped like B / C.  fp32: always honoured.  u's 16-bit dtype (e.g. column
    windows of the x_proj gradient operand): honoured when the launch takes the kernel that sums dB / dC over the d-tiles in a
    fixed order and writes the finished values once (segm_selective_scan_bwd_deterministic) - the result then has
    `dbc_native` True and dB / dC ARE the caller's tensors; otherwise fp32 tensors are allocated and `dbc_native` is False (the
    caller copies, as the reference's final cast does, selective_scan.cpp:488)."""
    if chunk and os.environ.get("SEGM_BWD_CHUNK"):        # experiments only: the backward's own chunking (the checkpoints are per 8 steps, not per chunk)
        chunk = int(os.environ["SEGM_BWD_CHUNK"])
    batch, seqlen, dim, dstate, groups, B4, C4 = _fill_scan_args(
        a.f, u, delta, A, B, C, D, z, delta_bias, delta_softplus, channel_last, time_order, nslices, chunk)
    if chunk == 0:
        raise RuntimeError("scan_bwd needs a chunk length (the forward's, or segm_selective_scan_bwd_default_chunk)")
    if ckpt is None:
        raise RuntimeError("scan_bwd needs the forward checkpoints (run the forward with need_ckpt=True)")
    if z is not None and out is None:
        raise RuntimeError("scan_bwd needs the forward's un-gated output when z is given")
    _check_seq("dout", dout, u, u.dtype)
    _check_seq("out", out, u, u.dtype)
    dev = u.device
    du = torch.empty_like(u, memory_format=torch.contiguous_format) if du is None else du
    ddelta = torch.empty_like(u, memory_format=torch.contiguous_format) if ddelta is None else ddelta
    if z is not None and dz is None:
        dz = torch.empty_like(u, memory_format=torch.contiguous_format)
    for name, t in (("du", du), ("ddelta", ddelta), ("dz", dz)):
        _check_seq(name, t, u, u.dtype)
    dA = torch.empty(dim, dstate, dtype=torch.float32, device=dev)
    a.f.out = L.seq_view(out, channel_last)
    a.f.ckpt = ckpt.data_ptr()
    native = False
    if dB is not None:
        dB, dC = _bc4(dB, channel_last), _bc4(dC, channel_last)
        native = dB.dtype == u.dtype and u.dtype != torch.float32
        for name, t in (("dB", dB), ("dC", dC)):
            if tuple(t.shape) != tuple(B4.shape) or t.dtype != (u.dtype if native else torch.float32) or t.device != dev:
                raise RuntimeError(f"{name} must be an fp32 (or, both, {u.dtype}) tensor of B's shape {tuple(B4.shape)} on the same device")
        if native and not lib.dll.segm_selective_scan_bwd_deterministic(a):
            dB, native = None, False                        # the atomically accumulating kernels need zeroed fp32 buffers
    if dB is None:
        dB = torch.empty(B4.shape, dtype=torch.float32, device=dev)
        dC = torch.empty(C4.shape, dtype=torch.float32, device=dev)
    a.dbc_native = int(native)
    dD = torch.empty(dim, dtype=torch.float32, device=dev) if D is not None else None
    ddb = torch.empty(dim, dtype=torch.float32, device=dev) if delta_bias is not None else None
    ws_bytes = lib.dll.segm_selective_scan_bwd_workspace_bytes(batch, dim, dstate, seqlen, chunk)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a.dout, a.du, a.ddelta, a.dz = (L.seq_view(t, channel_last) for t in (dout, du, ddelta, dz))
    a.dB, a.dC = L.bc_view(dB, channel_last), L.bc_view(dC, channel_last)
    a.dA, a.dD, a.ddelta_bias = dA.data_ptr(), L.fptr(dD), L.fptr(ddb)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    if B.dim() == 3:
        dB = dB.squeeze(2 if channel_last else 1)
    if C.dim() == 3:
        dC = dC.squeeze(2 if channel_last else 1)
    return dict(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC, dD=dD, ddelta_bias=ddb, dz=dz, dbc_native=native, _ws=ws)


# ---------------------------------------------------------------------------------------------------------
# causal depthwise conv1d
# ---------------------------------------------------------------------------------------------------------
def _fill_conv_args(a: L.Conv1dArgs, x, weight, bias, silu, channel_last, time_order, nslices):
    batch, seqlen, dim = _shape_bld(x, channel_last)
    if weight.dim() != 2 or weight.shape[0] != dim:
        raise RuntimeError(f"weight must be (dim={dim}, width)")
    width = weight.shape[1]
    if not 2 <= width <= 4:
        raise RuntimeError("causal_conv1d only supports width between 2 and 4")
    _check_vec("weight", weight.reshape(-1) if weight.is_contiguous() else weight, dim * width, x.device)
    _check_vec("bias", bias, dim, x.device)
    if time_order == L.TIME_INTERLEAVED and (nslices <= 0 or seqlen % nslices != 0):
        raise RuntimeError(f"seqlen {seqlen} must be divisible by nslices {nslices}")
    a.batch, a.dim, a.width, a.silu, a.seqlen = batch, dim, width, int(bool(silu)), seqlen
    a.dtype = L.dtype_code(x)
    a.time_order, a.nslices = int(time_order), int(nslices)
    a.x = L.seq_view(x, channel_last)
    a.weight, a.bias = weight.data_ptr(), L.fptr(bias)
    a.stream = L.stream_handle(x)
    return batch, seqlen, dim, width


def conv1d_fwd(lib: L.SegmLib, x, weight, bias=None, silu=False, *, channel_last=False,
               time_order=L.TIME_FORWARD, nslices=1, out=None):
    a = L.Conv1dArgs()
    _fill_conv_args(a, x, weight, bias, silu, channel_last, time_order, nslices)
    out = torch.empty_like(x, memory_format=torch.contiguous_format) if out is None else out
    _check_seq("out", out, x, x.dtype)
    a.out = L.seq_view(out, channel_last)
    lib.check(lib.dll.segm_causal_conv1d_fwd(a), "causal_conv1d_fwd")
    return out


def conv1d_fwd_multi(lib: L.SegmLib, calls):
    """several conv1d_fwd launches in one C call (`calls`: keyword dicts of conv1d_fwd with keys x, weight, bias, silu,
    channel_last, time_order, nslices, out): the three directions of a Mamba v3 layer run as one grid -> list of outputs"""
    n = len(calls)
    arr = (L.Conv1dArgs * n)()
    outs = []
    for i, c in enumerate(calls):
        x = c["x"]
        _fill_conv_args(arr[i], x, c["weight"], c.get("bias"), c.get("silu", False), c.get("channel_last", False),
                        c.get("time_order", L.TIME_FORWARD), c.get("nslices", 1))
        out = c.get("out")
        out = torch.empty_like(x, memory_format=torch.contiguous_format) if out is None else out
        _check_seq("out", out, x, x.dtype)
        arr[i].out = L.seq_view(out, c.get("channel_last", False))
        outs.append(out)
    lib.check(lib.dll.segm_causal_conv1d_fwd_multi(arr, n), "causal_conv1d_fwd_multi")
    return outs


def conv1d_bwd_multi(lib: L.SegmLib, calls):
    """the backward counterpart: keyword dicts of conv1d_bwd (x, weight, bias, dout, silu, channel_last, time_order, nslices, dx)
    -> list of (dx, dweight, dbias)"""
    n = len(calls)
    arr = (L.Conv1dArgs * n)()
    res, keep = [], []
    for i, c in enumerate(calls):
        r = _conv1d_bwd_prepare(lib, arr[i], c["x"], c["weight"], c.get("bias"), c["dout"], c.get("silu", False),
                                channel_last=c.get("channel_last", False), time_order=c.get("time_order", L.TIME_FORWARD),
                                nslices=c.get("nslices", 1), dx=c.get("dx"))
        res.append(r[:3])
        keep.append(r[3])
    lib.check(lib.dll.segm_causal_conv1d_bwd_multi(arr, n), "causal_conv1d_bwd_multi")
    return res


def conv1d_bwd(lib: L.SegmLib, x, weight, bias, dout, silu=False, *, channel_last=False,
               time_order=L.TIME_FORWARD, nslices=1, dx=None):
    """-> (dx, dweight fp32 (dim, width), dbias fp32 (dim) or None)"""
    a = L.Conv1dArgs()
    dx, dweight, dbias, _ws = _conv1d_bwd_prepare(lib, a, x, weight, bias, dout, silu, channel_last=channel_last,
                                                  time_order=time_order, nslices=nslices, dx=dx)
    lib.check(lib.dll.segm_causal_conv1d_bwd(a), "causal_conv1d_bwd")
    return dx, dweight, dbias


def _conv1d_bwd_prepare(lib, a, x, weight, bias, dout, silu=False, *, channel_last=False, time_order=L.TIME_FORWARD, nslices=1,
                        dx=None):
    batch, seqlen, dim, width = _fill_conv_args(a, x, weight, bias, silu, channel_last, time_order, nslices)
    _check_seq("dout", dout, x, x.dtype)
    dx = torch.empty_like(x, memory_format=torch.contiguous_format) if dx is None else dx
    _check_seq("dx", dx, x, x.dtype)
    dev = x.device
    dweight = torch.empty(dim, width, dtype=torch.float32, device=dev)
    dbias = torch.empty(dim, dtype=torch.float32, device=dev) if bias is not None else None
    ws_bytes = lib.dll.segm_causal_conv1d_bwd_workspace_bytes(batch, dim, width, seqlen)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a.dout, a.dx = L.seq_view(dout, channel_last), L.seq_view(dx, channel_last)
    a.dweight, a.dbias = dweight.data_ptr(), L.fptr(dbias)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    return dx, dweight, dbias, ws


# ---------------------------------------------------------------------------------------------------------
# 3x3x3 convolution weight gradient (stem / decoder)
# ---------------------------------------------------------------------------------------------------------
def conv3d_k3_wgrad_supported(x: torch.Tensor, dy: torch.Tensor) -> bool:
    """Shapes / layouts the MFMA weight-gradient kernel takes (everything else stays on MIOpen)."""
    if x.dim() != 5 or dy.dim() != 5 or x.dtype not in (torch.bfloat16, torch.float16) or dy.dtype != x.dtype:
        return False
    if x.shape[0] != dy.shape[0] or x.shape[2:] != dy.shape[2:]:
        return False
    if (x.shape[1] % 48 and x.shape[1] > 48) or dy.shape[1] % 48 or x.shape[4] % 8:
        return False
    for t in (x, dy):
        if t.stride(4) != 1 or any(t.stride(i) % 8 for i in range(4)) or t.data_ptr() % 16:
            return False
    return conv3d_k3_wgrad_spans_fit(x, dy)


def conv3d_k3_wgrad_spans_fit(x: torch.Tensor, dy: torch.Tensor) -> bool:
    """The weight-gradient kernel addresses a (batch, plane, 48-channel block) through 32-bit byte offsets: 48 channel strides and
    a plane's rows must stay below the buffer descriptor's 0xFFFFF000 bytes - the same bound segm_conv3d_k3_wgrad checks
    (csrc/conv3d_wgrad.hip, `fits32`; volumes up to ~350^3 elements per channel).  Beyond it the routing falls back to the
    vendor route instead of raising."""
    lim = 0xFFFFF000 - 4096
    W, H = x.shape[4], x.shape[3]
    for t in (x, dy):
        if t.stride(1) * 96 + 2 * W >= lim or t.stride(3) * 2 * H >= lim:
            return False
    return True


def conv3d_k3_wgrad(lib: L.SegmLib, x: torch.Tensor, dy: torch.Tensor, out_dtype=torch.bfloat16) -> torch.Tensor:
    """dW (Cout, Cin, 3, 3, 3) of a stride-1 pad-1 3x3x3 convolution; x (B, Cin, D, H, W), dy (B, Cout, D, H, W), bf16."""
    if not conv3d_k3_wgrad_supported(x, dy):
        raise RuntimeError("conv3d_k3_wgrad: unsupported shape / dtype / layout")
    B, cin, D, H, W = x.shape
    cout = dy.shape[1]
    a = L.Conv3dWgradArgs()
    a.batch, a.cin, a.cout, a.depth, a.height, a.width = B, cin, cout, D, H, W
    a.dtype = L.dtype_code(x)
    a.dw_dtype = L.dtype_code(torch.empty(0, dtype=out_dtype))
    a.x, a.dy = x.data_ptr(), dy.data_ptr()
    a.x_stride_b, a.x_stride_c, a.x_stride_z, a.x_stride_y = x.stride()[:4]
    a.dy_stride_b, a.dy_stride_c, a.dy_stride_z, a.dy_stride_y = dy.stride()[:4]
    dw = torch.empty(cout, cin, 3, 3, 3, dtype=out_dtype, device=x.device)
    ws_bytes = lib.dll.segm_conv3d_k3_wgrad_workspace_bytes(B, cin, cout, D, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    a.dw, a.workspace, a.workspace_bytes = dw.data_ptr(), ws.data_ptr(), ws_bytes
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_conv3d_k3_wgrad(a), "conv3d_k3_wgrad")
    return dw


# ---------------------------------------------------------------------------------------------------------
# 3x3x3 convolution forward / data gradient (48 input channels)
# ---------------------------------------------------------------------------------------------------------
def pack_conv3d_weight(weight: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """(Cout, Cin <= 48, 3, 3, 3) -> (Cout, 3, 3, 3, 48) contiguous in the activations' 16-bit dtype: the layout
    segm_conv3d_k3_fwd keeps in registers."""
    w = weight.permute(0, 2, 3, 4, 1)
    if w.shape[-1] < 48:                                   # a narrow first layer: zero input channels up to 48
        w = torch.nn.functional.pad(w, (0, 48 - w.shape[-1]))
    return w.contiguous().to(dtype)


def pack_conv3d_weight_for_dgrad(weight: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """The data gradient of a stride-1 pad-1 3x3x3 convolution is the forward convolution of dy with the spatially
    flipped, channel-transposed weights: (Cout, Cin, 3, 3, 3) -> packed (Cin, 3, 3, 3, Cout)."""
    return pack_conv3d_weight(weight.flip(2, 3, 4).transpose(0, 1), dtype)


def conv3d_k3_fwd_supported(x: torch.Tensor, cout: int) -> bool:
    if x.dim() != 5 or x.dtype not in (torch.bfloat16, torch.float16) or not 1 <= x.shape[1] <= 48 or cout % 16 or x.shape[4] % 8:
        return False
    return x.stride(4) == 1 and not any(x.stride(i) % 8 for i in range(4)) and x.data_ptr() % 16 == 0


CONV_FWD_ACCUMULATE, CONV_FWD_CHAIN, CONV_FWD_PITCH48, CONV_FWD_CHAIN32 = 1, 2, 4, 8     # segm_conv_fwd_flags


def conv3d_k3_fwd(lib: L.SegmLib, x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, accumulate: bool = False, chain: bool = False,
                  pitch48: bool = False, chain32: bool = False, want_stats: bool = False):
    """y (B, Cout, D, H, W) = conv3d(x, w, bias, stride 1, padding 1); x (B, <= 48, D, H, W) bf16 / fp16, w_packed from
    pack_conv3d_weight().  `out` + `accumulate`: add to an existing result (the next 48-channel block of a wider
    layer; Cout % 48 == 0).  `chain`: the pipelined-K-parts kernel (Cout % 48 == 0); `pitch48`: its unpadded LDS layout; `chain32`: the
    same pipeline on 32-wide x blocks, two workgroups per CU (exclusive with `chain`).
    want_stats (chain + pitch48, or chain32): -> (y, stats), stats = fp32 (B, Cout, nparts, 4) partial {count, sum, sum of squares, -}
    of what this launch stored, for instnorm_fwd(..., stats=stats); (y, None) when the launch has no statistics epilogue."""
    cout = w_packed.shape[0]
    if not conv3d_k3_fwd_supported(x, cout):
        raise RuntimeError("conv3d_k3_fwd: unsupported shape / dtype / layout")
    if tuple(w_packed.shape[1:]) != (3, 3, 3, 48) or w_packed.dtype != x.dtype or not w_packed.is_contiguous():
        raise RuntimeError("conv3d_k3_fwd: w_packed must be a contiguous (Cout, 3, 3, 3, 48) tensor of x's dtype")
    if (accumulate or chain or chain32) and cout % 48:
        raise RuntimeError("conv3d_k3_fwd: accumulate / chain need Cout % 48 == 0")
    if chain32 and (chain or pitch48):
        raise RuntimeError("conv3d_k3_fwd: chain32 excludes chain / pitch48")
    B, _, D, H, W = x.shape
    if out is None:
        if accumulate:
            raise RuntimeError("conv3d_k3_fwd: accumulate needs `out`")
        y = volume_empty(B, cout, (D, H, W), x.dtype, x.device)
    else:
        y = out
        if tuple(y.shape) != (B, cout, D, H, W) or y.dtype != x.dtype or y.device != x.device or y.stride(4) != 1 or \
                any(y.stride(i) % 8 for i in range(4)) or y.data_ptr() % 16:
            raise RuntimeError("conv3d_k3_fwd: `out` must be a (B, Cout, D, H, W) tensor of x's dtype, W contiguous, 16-byte rows")
    if bias is not None:
        bias = bias.float().contiguous()
    a = L.Conv3dFwdArgs()
    a.batch, a.cin, a.cout, a.depth, a.height, a.width = B, x.shape[1], cout, D, H, W
    a.dtype = L.dtype_code(x)
    if pitch48 and not chain:
        raise RuntimeError("conv3d_k3_fwd: pitch48 is a variant of the chained kernel")
    a.flags = (CONV_FWD_ACCUMULATE if accumulate else 0) | (CONV_FWD_CHAIN if chain else 0) | (CONV_FWD_PITCH48 if pitch48 else 0) | \
        (CONV_FWD_CHAIN32 if chain32 else 0)
    a.x, a.y, a.w_packed = x.data_ptr(), y.data_ptr(), w_packed.data_ptr()
    a.x_stride_b, a.x_stride_c, a.x_stride_z, a.x_stride_y = x.stride()[:4]
    a.y_stride_b, a.y_stride_c, a.y_stride_z, a.y_stride_y = y.stride()[:4]
    a.bias = bias.data_ptr() if bias is not None else None
    a.stream = L.stream_handle(x)
    stats = None
    if want_stats and ((chain and pitch48) or chain32):
        nparts = lib.dll.segm_conv3d_k3_fwd_stats_parts(D, H, W, B, cout, a.flags)
        if nparts > 0:
            # (a workgroup whose y part is empty - ysplit >= 16 with H not a multiple of it - writes a zero-count slot: ADVICE r05)
            stats = torch.empty(B, cout, nparts, 4, dtype=torch.float32, device=x.device)
            a.stats_partials, a.stats_nparts = stats.data_ptr(), nparts
    lib.check(lib.dll.segm_conv3d_k3_fwd(a), "conv3d_k3_fwd")
    return (y, stats) if want_stats else y


# ---------------------------------------------------------------------------------------------------------
# channel-last 3x3x3 convolution (round 6 prototype, csrc/conv3d_cl.hip)
CONV_CL_ACCUMULATE, CONV_CL_WAVES8 = 1, 2
_cl_index_cache = {}


def conv3d_cl_weight_image(lib: L.SegmLib, w: torch.Tensor, dtype=None) -> torch.Tensor:
    """(48, 48, 3, 3, 3) weights -> the (14, 9, 64, 8) image of MFMA operand fragments segm_conv3d_k3_fwd_cl keeps in LDS (one
    gather through the index map the library exports).  For the data gradient pass flip(w, (2, 3, 4)).transpose(0, 1)."""
    if tuple(w.shape) != (48, 48, 3, 3, 3):
        raise RuntimeError("conv3d_cl_weight_image: weights must be (48, 48, 3, 3, 3)")
    key = str(w.device)
    idx = _cl_index_cache.get(key)
    if idx is None:
        n = 14 * 9 * 64 * 8
        host = torch.empty(n, dtype=torch.int32)
        lib.check(lib.dll.segm_conv3d_k3_cl_pack_index(host.data_ptr(), n), "conv3d_k3_cl_pack_index")
        idx = host.to(w.device).long()
        _cl_index_cache[key] = idx
    flat = torch.cat([w.reshape(-1), w.new_zeros(1)])       # index -1 -> the appended zero
    return flat[idx].view(14, 9, 64, 8).to(dtype or w.dtype).contiguous()


def conv3d_k3_fwd_cl(lib: L.SegmLib, x: torch.Tensor, w_image: torch.Tensor, bias=None, out=None, accumulate=False, waves8=False):
    """y (B, D, H, W, 48) = conv3d(x (B, D, H, W, 48) channel-last, 3x3x3, stride 1, pad 1); bf16 / fp16."""
    if x.dim() != 5 or x.shape[4] != 48 or x.stride(4) != 1 or x.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("conv3d_k3_fwd_cl: x must be a (B, D, H, W, 48) 16-bit tensor with contiguous channels")
    B, D, H, W, _ = x.shape
    y = out if out is not None else torch.empty(B, D, H, W, 48, dtype=x.dtype, device=x.device)
    if accumulate and out is None:
        raise RuntimeError("conv3d_k3_fwd_cl: accumulate needs `out`")
    a = L.Conv3dClArgs()
    a.batch, a.channels, a.depth, a.height, a.width = B, 48, D, H, W
    a.dtype = L.dtype_code(x)
    a.flags = (CONV_CL_ACCUMULATE if accumulate else 0) | (CONV_CL_WAVES8 if waves8 else 0)
    a.x, a.y, a.w_image = x.data_ptr(), y.data_ptr(), w_image.data_ptr()
    a.x_stride_b, a.x_stride_z, a.x_stride_y, a.x_stride_x = x.stride()[:4]
    a.y_stride_b, a.y_stride_z, a.y_stride_y, a.y_stride_x = y.stride()[:4]
    if bias is not None:
        bias = bias.float().contiguous()
    a.bias = bias.data_ptr() if bias is not None else None
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_conv3d_k3_fwd_cl(a), "conv3d_k3_fwd_cl")
    return y


# ---------------------------------------------------------------------------------------------------------
# 3x3x3 convolution of wide layers on small volumes (csrc/conv3d_cube.hip, ABI 10)
# ---------------------------------------------------------------------------------------------------------
CONV_CUBE_ACCUMULATE = 1
_cube_index_cache = {}


def conv3d_cube_supported(x: torch.Tensor, cout: int) -> bool:
    """does segm_conv3d_k3_cube_fwd take this input (B, Cin, D, H, W) for `cout` output channels"""
    if x.dim() != 5 or x.dtype not in (torch.bfloat16, torch.float16) or x.stride(4) != 1:
        return False
    B, C, D, H, W = x.shape
    if C % 32 != 0 or (cout % 64 != 0 and cout % 96 != 0) or D % 8 or H % 8 or W % 8 or min(B, C, D, H, W) <= 0:
        return False
    return all(s % 8 == 0 and s > 0 for s in x.stride()[:4]) and x.data_ptr() % 16 == 0


def conv3d_cube_index(lib: L.SegmLib, cout_w: int, cin_w: int, flipped: bool, device) -> torch.Tensor:
    """int32 map: element i of the kernel's weight image <- flat index into the (cout_w, cin_w, 3, 3, 3) weight"""
    key = (cout_w, cin_w, bool(flipped), str(device))
    idx = _cube_index_cache.get(key)
    if idx is None:
        n = cout_w * cin_w * 27
        host = torch.empty(n, dtype=torch.int32)
        lib.check(lib.dll.segm_conv3d_k3_cube_pack_index(host.data_ptr(), n, cout_w, cin_w, 1 if flipped else 0), "conv3d_k3_cube_pack_index")
        idx = host.to(device)
        _cube_index_cache[key] = idx
    return idx


_cube_desc_cache = {}


def conv3d_cube_weight_image(lib: L.SegmLib, w: torch.Tensor, flipped: bool = False, dtype=None, by_index: bool = False) -> torch.Tensor:
    """(Cout, Cin, 3, 3, 3) weights (or a channel slice of a wider weight) -> the fragment image segm_conv3d_k3_cube_fwd streams
    (`flipped`: the image of the data gradient, a convolution of dy with flip(w)^T).  On the device one launch of the pack kernel
    (segm_conv3d_k3_cube_pack_multi with a single cached descriptor); `by_index` (and layouts the kernel does not take): a gather
    through the index map the library exports - the definition the pack kernel is tested against."""
    direct = (not by_index and L.on_device(w) and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and tuple(w.stride()[1:]) == (27, 9, 3, 1) and
              w.dtype in (torch.bfloat16, torch.float16) and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and 0 < w.stride(0) < (1 << 31) and
              (dtype is None or dtype == w.dtype))
    if direct:
        key = (w.shape[0], w.shape[1], w.stride(0), bool(flipped), str(w.device))
        hit = _cube_desc_cache.get(key)
        if hit is None:
            hit = cube_pack_descs([(0, 0, w.shape[0], w.shape[1], w.stride(0), flipped)], w.device)
            _cube_desc_cache[key] = hit
        out = torch.empty(w.shape[0] * w.shape[1] * 27, dtype=w.dtype, device=w.device)
        lib.check(lib.dll.segm_conv3d_k3_cube_pack_multi(w.data_ptr(), out.data_ptr(), hit[0].data_ptr(), 1, hit[1], L.stream_handle(out)),
                  "conv3d_k3_cube_pack_multi")
        return out
    idx = conv3d_cube_index(lib, w.shape[0], w.shape[1], flipped, w.device)
    img = torch.index_select(w.reshape(-1), 0, idx)
    return img if dtype is None or dtype == img.dtype else img.to(dtype)


def cube_pack_descs(items, device) -> tuple:
    """items: [(src_off, out_off, cout_w, cin_w, co_stride, flipped)] -> (int32 (n, 8) device tensor laid out as
    segm_cube_pack_desc, total blocks) for conv3d_cube_pack_multi"""
    rows, first = [], 0
    for src_off, out_off, cout_w, cin_w, co_stride, flipped in items:
        rows.append([src_off & 0xffffffff, src_off >> 32, out_off & 0xffffffff, out_off >> 32, cout_w, cin_w, co_stride, int(bool(flipped)), first, 0])
        first += (cin_w // 16) * (cout_w // 32) if flipped else (cout_w // 16) * (cin_w // 32)
    t = torch.tensor(rows, dtype=torch.int64).to(torch.int32).to(device).contiguous()
    return t, first


def conv3d_cube_pack_multi(lib: L.SegmLib, src: torch.Tensor, out: torch.Tensor, descs: torch.Tensor, nblocks: int) -> torch.Tensor:
    """every weight image described by `descs` (cube_pack_descs) from the 16-bit weight buffer `src` into `out`, one launch"""
    if src.dtype not in (torch.bfloat16, torch.float16) or out.dtype != src.dtype or descs.dtype != torch.int32 or descs.dim() != 2 or \
            descs.shape[1] != 10 or not (src.is_contiguous() and out.is_contiguous() and descs.is_contiguous()) or out.data_ptr() & 15:
        raise RuntimeError("conv3d_cube_pack_multi: 16-bit src / out, int32 (n, 10) descriptors")
    lib.check(lib.dll.segm_conv3d_k3_cube_pack_multi(src.data_ptr(), out.data_ptr(), descs.data_ptr(), descs.shape[0], nblocks,
                                                     L.stream_handle(out)), "conv3d_k3_cube_pack_multi")
    return out


def conv3d_cube_plan(lib: L.SegmLib, B, cin, cout, D, H, W, nt=0, splits=0):
    c_nt, c_s, c_ws = C.c_int32(nt), C.c_int32(splits), C.c_int64(0)
    lib.check(lib.dll.segm_conv3d_k3_cube_plan(B, cin, cout, D, H, W, C.byref(c_nt), C.byref(c_s), C.byref(c_ws)), "conv3d_k3_cube_plan")
    return c_nt.value, c_s.value, c_ws.value


def conv3d_k3_cube_fwd(lib: L.SegmLib, x: torch.Tensor, w_image: torch.Tensor, cout: int, bias=None, out=None, accumulate=False,
                       nt=0, splits=0, want_stats=False):
    """y (B, cout, D, H, W) = conv3d(x (B, Cin, D, H, W), 3x3x3, stride 1, pad 1) through the cube kernel; bf16 / fp16.  The
    fp32 partial sums live in a per-device scratch buffer that grows to the largest layer seen (stream-ordered reuse).
    want_stats: -> (y, stats), stats = fp32 (B, cout, nparts, 4) partial {count, sum, sum of squares, -} of what the launch stored
    (per wave of the storing launch: 128 or 512 voxels), for instnorm_fwd(..., stats=stats)."""
    if not conv3d_cube_supported(x, cout):
        raise RuntimeError("conv3d_k3_cube_fwd: unsupported input")
    B, cin, D, H, W = x.shape
    if accumulate and out is None:
        raise RuntimeError("conv3d_k3_cube_fwd: accumulate needs `out`")
    y = out if out is not None else torch.empty(B, cout, D, H, W, dtype=x.dtype, device=x.device)
    nt, splits, need = conv3d_cube_plan(lib, B, cin, cout, D, H, W, nt, splits)
    # the partial sums of this call (none with one split): a fresh tensor per call - the caching allocator hands the same block back
    # call after call on one stream, and a buffer shared between calls would be a race as soon as two streams convolve
    ws = torch.empty(need, dtype=torch.float32, device=x.device) if need > 0 else None
    a = L.Conv3dCubeArgs()
    a.batch, a.cin, a.cout, a.depth, a.height, a.width = B, cin, cout, D, H, W
    a.dtype = L.dtype_code(x)
    a.flags = CONV_CUBE_ACCUMULATE if accumulate else 0
    a.nt, a.splits = nt, splits
    a.x, a.y, a.w_image = x.data_ptr(), y.data_ptr(), w_image.data_ptr()
    a.x_stride_b, a.x_stride_c, a.x_stride_z, a.x_stride_y = x.stride()[:4]
    a.y_stride_b, a.y_stride_c, a.y_stride_z, a.y_stride_y = y.stride()[:4]
    if bias is not None:
        bias = bias.float().contiguous()
    a.bias = bias.data_ptr() if bias is not None else None
    a.workspace, a.workspace_elems = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
    a.stream = L.stream_handle(x)
    stats = None
    if want_stats:
        stats = torch.empty(B, cout, lib.dll.segm_conv3d_k3_cube_stats_parts(D, H, W, splits), 4, dtype=torch.float32, device=x.device)
        a.stats_partials, a.stats_nparts = stats.data_ptr(), stats.shape[2]
    lib.check(lib.dll.segm_conv3d_k3_cube_fwd(a), "conv3d_k3_cube_fwd")
    return (y, stats) if want_stats else y


def conv3d_cube_wgrad_supported(x: torch.Tensor, dy: torch.Tensor) -> bool:
    if x.dim() != 5 or dy.dim() != 5 or x.dtype not in (torch.bfloat16, torch.float16) or dy.dtype != x.dtype:
        return False
    if x.shape[0] != dy.shape[0] or x.shape[2:] != dy.shape[2:] or x.shape[1] % 32 or dy.shape[1] % 64 or x.shape[2] % 8 or x.shape[3] % 8 or x.shape[4] % 8:
        return False
    for t in (x, dy):
        if t.stride(4) != 1 or any(t.stride(i) % 8 or t.stride(i) <= 0 for i in range(4)) or t.data_ptr() % 16:
            return False
    return True


def conv3d_k3_cube_wgrad(lib: L.SegmLib, x: torch.Tensor, dy: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    """dW (Cout, Cin, 3, 3, 3) of a stride-1 pad-1 3x3x3 convolution through segm_conv3d_k3_cube_wgrad; x (B, Cin, D, H, W),
    dy (B, Cout, D, H, W), bf16 / fp16"""
    if not conv3d_cube_wgrad_supported(x, dy):
        raise RuntimeError("conv3d_k3_cube_wgrad: unsupported shape / dtype / layout")
    B, cin, D, H, W = x.shape
    cout = dy.shape[1]
    a = L.Conv3dWgradArgs()
    a.batch, a.cin, a.cout, a.depth, a.height, a.width = B, cin, cout, D, H, W
    a.dtype = L.dtype_code(x)
    a.dw_dtype = L.dtype_code(torch.empty(0, dtype=out_dtype))
    a.x, a.dy = x.data_ptr(), dy.data_ptr()
    a.x_stride_b, a.x_stride_c, a.x_stride_z, a.x_stride_y = x.stride()[:4]
    a.dy_stride_b, a.dy_stride_c, a.dy_stride_z, a.dy_stride_y = dy.stride()[:4]
    dw = torch.empty(cout, cin, 3, 3, 3, dtype=out_dtype, device=x.device)
    ws_bytes = lib.dll.segm_conv3d_k3_cube_wgrad_workspace_bytes(B, cin, cout, D, H, W)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=x.device)
    a.dw, a.workspace, a.workspace_bytes = dw.data_ptr(), ws.data_ptr(), ws.numel() * 4
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_conv3d_k3_cube_wgrad(a), "conv3d_k3_cube_wgrad")
    return dw


# ---------------------------------------------------------------------------------------------------------
# InstanceNorm3d (+ residual) (+ activation)
# ---------------------------------------------------------------------------------------------------------
ACT_CODES = {"none": 0, "relu": 1, "leaky_relu": 2}


# ---------------------------------------------------------------------------------------------------------
# volumes with a padded channel stride
# ---------------------------------------------------------------------------------------------------------
# A dense (B, C, 128, 128, 128) 16-bit volume has a channel stride of exactly 4 MiB: the C lines a convolution writes for one
# output row (one per channel, same z / y / x) all fall into the same L2 set and the same memory channel.  Measured on MI355X
# (profiles/r02_conv_stride_pad.log): the 48 -> 48 3x3x3 kernel takes 0.669 ms on dense tensors, 0.586 ms with 64 elements of
# padding per channel of the OUTPUT and 0.530 ms with input and output padded; the 1x1x1 kernel 0.284 -> 0.247 ms; 64^3 volumes
# (0.5 MiB) do not care.  The convolution kernels therefore allocate such outputs with `volume_empty`; their consumers take the
# strides (convolutions, the instance-norm kernels) or see an ordinary strided tensor (ATen).
_VOLUME_PAD = 192                 # elements: 384 bytes = three 128-byte lines
_VOLUME_PAD_ON = os.environ.get("SEGM_VOLUME_PAD", "1") == "1"
_VOLUME_PAD_QUANTUM = 1 << 20     # channel sizes (bytes) that are multiples of this get the padding (tests lower it)


def volume_empty(B: int, C: int, spatial, dtype: torch.dtype, device) -> torch.Tensor:
    """(B, C, *spatial) tensor, dense inside a channel; the channel stride is padded by _VOLUME_PAD elements when the dense one
    would be a multiple of 1 MiB (C >= 16: with a handful of channels nothing aliases; 16-bit dtypes only: fp32 volumes go to
    vendor kernels that would copy a strided input)."""
    S = 1
    for d in spatial:
        S *= int(d)
    nbytes = S * torch.empty(0, dtype=dtype).element_size()
    if not (_VOLUME_PAD_ON and C >= 16 and dtype in (torch.bfloat16, torch.float16) and nbytes >= _VOLUME_PAD_QUANTUM
            and nbytes % _VOLUME_PAD_QUANTUM == 0):
        return torch.empty(B, C, *spatial, dtype=dtype, device=device)
    buf = torch.empty(B, C, S + _VOLUME_PAD, dtype=dtype, device=device)
    return buf[:, :, :S].view(B, C, *spatial)


def channel_dense(x: torch.Tensor) -> bool:
    """(B, C, *spatial) with dense channels (contiguous spatial dims) and batch stride = C * channel stride: what the kernels
    that take a channel / instance stride accept (a contiguous tensor, or one from volume_empty)."""
    if x.dim() < 3:
        return False
    s = 1
    for i in range(x.dim() - 1, 1, -1):
        if x.shape[i] != 1 and x.stride(i) != s:
            return False
        s *= x.shape[i]
    if x.shape[1] == 1:                                   # one channel: the batch stride is the instance stride
        return x.shape[0] == 1 or x.stride(0) >= s
    return x.stride(1) >= s and (x.shape[0] == 1 or x.stride(0) == x.shape[1] * x.stride(1))


def instance_stride(x: torch.Tensor) -> int:
    """elements between consecutive (b, c) instances of a channel_dense tensor (strides of size-1 dimensions mean nothing)"""
    if x.shape[1] > 1:
        return x.stride(1)
    if x.shape[0] > 1:
        return x.stride(0)
    return x[0, 0].numel()


def _norm_geom(x):
    if x.dim() < 3 or not channel_dense(x):
        raise RuntimeError("instnorm: x must be a (B, C, *spatial) tensor with dense channels (contiguous, or a padded channel stride)")
    inst = x.shape[0] * x.shape[1]
    return inst, x.numel() // inst


def instnorm_fwd(lib: L.SegmLib, x, residual=None, act="none", slope=0.01, eps=1e-5, stats=None):
    """-> (y, mean, rstd): y = act(IN(x) + residual); mean / rstd fp32 (B * C).
    stats: fp32 (B, C, nparts, 4) of {count, sum, sum of squares, -} partials summed by the producer of x (conv3d_k3_fwd's `stats`):
    the statistics pass over x is skipped."""
    inst, S = _norm_geom(x)
    if stats is not None and (stats.dtype != torch.float32 or stats.dim() != 4 or stats.shape[0] * stats.shape[1] != inst or
                              stats.shape[3] != 4 or not stats.is_contiguous() or stats.device != x.device):
        raise RuntimeError("instnorm: stats must be a contiguous fp32 (B, C, nparts, 4) tensor on x's device")
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or not channel_dense(residual)):
        raise RuntimeError("instnorm: residual must match x (shape, dtype, dense channels)")
    a = L.InstNormFwdArgs()
    a.instances, a.dtype, a.act, a.spatial = inst, L.dtype_code(x), ACT_CODES[act], S
    a.slope, a.eps = float(slope), float(eps)
    y = volume_empty(x.shape[0], x.shape[1], x.shape[2:], x.dtype, x.device)
    a.x_instance_stride, a.y_instance_stride = instance_stride(x), instance_stride(y)
    a.residual_instance_stride = instance_stride(residual) if residual is not None else 0
    mean = torch.empty(inst, dtype=torch.float32, device=x.device)
    rstd = torch.empty(inst, dtype=torch.float32, device=x.device)
    ws_bytes = lib.dll.segm_instnorm_workspace_bytes(inst, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    a.x, a.residual, a.y = x.data_ptr(), (residual.data_ptr() if residual is not None else None), y.data_ptr()
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.workspace, a.workspace_bytes, a.stream = ws.data_ptr(), ws_bytes, L.stream_handle(x)
    if stats is not None:
        a.stats_partials, a.stats_nparts = stats.data_ptr(), stats.shape[2]
    lib.check(lib.dll.segm_instnorm_fwd(a), "instnorm_fwd")
    return y, mean, rstd


def instnorm_bwd(lib: L.SegmLib, x, dy, mean, rstd, y=None, act="none", slope=0.01, want_dresidual=False):
    """-> (dx, dresidual or None).  `y` is needed iff act != none and the forward added a residual."""
    inst, S = _norm_geom(x)
    if dy.shape != x.shape or dy.dtype != x.dtype:
        raise RuntimeError("instnorm: dy must match x")
    if not channel_dense(dy):
        dy = dy.contiguous()
    a = L.InstNormBwdArgs()
    a.instances, a.dtype, a.act, a.spatial = inst, L.dtype_code(x), ACT_CODES[act], S
    a.slope = float(slope)
    if y is not None and (y.shape != x.shape or y.dtype != x.dtype or not channel_dense(y)):
        raise RuntimeError("instnorm: y must match x (shape, dtype, dense channels)")
    dx = volume_empty(x.shape[0], x.shape[1], x.shape[2:], x.dtype, x.device)
    dres = volume_empty(x.shape[0], x.shape[1], x.shape[2:], x.dtype, x.device) if want_dresidual else None
    a.x_instance_stride, a.dy_instance_stride, a.dx_instance_stride = instance_stride(x), instance_stride(dy), instance_stride(dx)
    a.y_instance_stride = instance_stride(y) if y is not None else 0
    a.dresidual_instance_stride = instance_stride(dres) if dres is not None else 0
    ws_bytes = lib.dll.segm_instnorm_workspace_bytes(inst, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    a.x, a.dy, a.y = x.data_ptr(), dy.data_ptr(), (y.data_ptr() if y is not None else None)
    a.mean, a.rstd, a.dx = mean.data_ptr(), rstd.data_ptr(), dx.data_ptr()
    a.dresidual = dres.data_ptr() if dres is not None else None
    a.workspace, a.workspace_bytes, a.stream = ws.data_ptr(), ws_bytes, L.stream_handle(x)
    lib.check(lib.dll.segm_instnorm_bwd(a), "instnorm_bwd")
    return dx, dres


# ---------------------------------------------------------------------------------------------------------
# depth-to-space / space-to-depth by 2 x 2 x 2 (the permute behind a kernel-2 stride-2 transposed convolution)
# ---------------------------------------------------------------------------------------------------------
def _d2s_shape_ok(t: torch.Tensor) -> bool:
    return t.dtype in (torch.bfloat16, torch.float16) and t.dim() == 5 and t.shape[4] % 8 == 0 and t.shape[1] % 8 == 0 and t.is_contiguous()


def depth_to_space2_supported(t: torch.Tensor) -> bool:
    return L.on_device(t) and _d2s_shape_ok(t)


def depth_to_space2(lib: L.SegmLib, blk: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """blk (B, C * 8, D, H, W) contiguous [channel index = ((c * 2 + i) * 2 + j) * 2 + k] -> vol (B, C, 2D, 2H, 2W) with
    vol[b, c, 2z+i, 2y+j, 2x+k] = blk[b, c, i, j, k, z, y, x]; `out`: a volume with unit x stride (e.g. ops_raw.volume_empty)."""
    B, C8, D, H, W = blk.shape
    if not _d2s_shape_ok(blk):
        raise RuntimeError("depth_to_space2: blk must be a contiguous 16-bit (B, C * 8, D, H, W) tensor with W % 8 == 0")
    Cc = C8 // 8
    if out is None:
        out = torch.empty(B, Cc, 2 * D, 2 * H, 2 * W, dtype=blk.dtype, device=blk.device)
    _d2s_call(lib, blk, out, 0)
    return out


def space_to_depth2(lib: L.SegmLib, vol: torch.Tensor) -> torch.Tensor:
    """the inverse gather: vol (B, C, 2D, 2H, 2W), unit x stride -> blk (B, C * 8, D, H, W) contiguous"""
    B, Cc, D2, H2, W2 = vol.shape
    if D2 % 2 or H2 % 2 or W2 % 16 or vol.dtype not in (torch.bfloat16, torch.float16) or vol.stride(4) != 1:
        raise RuntimeError("space_to_depth2: vol must be a 16-bit (B, C, 2D, 2H, 2W) tensor with unit x stride and 2W % 16 == 0")
    blk = torch.empty(B, Cc * 8, D2 // 2, H2 // 2, W2 // 2, dtype=vol.dtype, device=vol.device)
    _d2s_call(lib, blk, vol, 1)
    return blk


def _d2s_call(lib, blk, vol, direction):
    B, C8, D, H, W = blk.shape
    if tuple(vol.shape) != (B, C8 // 8, 2 * D, 2 * H, 2 * W) or vol.dtype != blk.dtype or vol.stride(4) != 1 or vol.device != blk.device:
        raise RuntimeError("depth_to_space2: vol must be (B, C, 2D, 2H, 2W) of blk's dtype with unit x stride")
    a = L.D2sArgs()
    a.batch, a.channels, a.depth, a.height, a.width = B, C8 // 8, D, H, W
    a.dtype, a.direction = L.dtype_code(blk), direction
    a.blk, a.vol = blk.data_ptr(), vol.data_ptr()
    a.vol_stride_b, a.vol_stride_c, a.vol_stride_z, a.vol_stride_y = vol.stride(0), vol.stride(1), vol.stride(2), vol.stride(3)
    a.stream = L.stream_handle(blk)
    lib.check(lib.dll.segm_depth_to_space2(a), "depth_to_space2")


# ---------------------------------------------------------------------------------------------------------
# channel-first <-> channel-last
# ---------------------------------------------------------------------------------------------------------
def add3_supported(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> bool:
    n = 4 if a.dtype == torch.float32 else 8
    return a.dtype in (torch.float32, torch.float16, torch.bfloat16) and a.shape == b.shape == c.shape and a.dtype == b.dtype == c.dtype \
        and a.is_contiguous() and b.is_contiguous() and c.is_contiguous() and a.numel() > 0 and a.numel() % n == 0 \
        and not ((a.data_ptr() | b.data_ptr() | c.data_ptr()) & 15)


def add3(lib: L.SegmLib, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + b + c in one pass (fp32 sum, one rounding); `out` may be `a`."""
    if not add3_supported(a, b, c):
        raise RuntimeError("add3: three dense 16-byte aligned tensors of one shape and dtype")
    out = torch.empty_like(a) if out is None else out
    if out.shape != a.shape or out.dtype != a.dtype or not out.is_contiguous() or out.data_ptr() & 15:
        raise RuntimeError("add3: `out` must match the operands")
    p = L.Add3Args()
    p.count, p.dtype = a.numel(), L.dtype_code(a)
    p.a, p.b, p.c, p.out = a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr()
    p.stream = L.stream_handle(a)
    lib.check(lib.dll.segm_add3(p), "add3")
    return out


def gather16_compact_map(idx: torch.Tensor) -> Optional[torch.Tensor]:
    """int32 (n,) source indices, n % 8 == 0 -> int32 (n / 8, 2) (first index, step) when every group of eight consecutive indices
    is an arithmetic progression (the packed weight layouts are: eight input channels of one (output channel, tap)), else None"""
    if idx.numel() == 0 or idx.numel() % 8:
        return None
    m = idx.view(-1, 8)
    step = m[:, 1] - m[:, 0]
    ar = torch.arange(8, dtype=idx.dtype, device=idx.device)
    if not bool((m == m[:, :1] + step[:, None] * ar).all()):
        return None
    return torch.stack([m[:, 0], step], 1).contiguous()


def gather16(lib: L.SegmLib, src: torch.Tensor, idx: torch.Tensor, out: torch.Tensor, compact: bool = False) -> torch.Tensor:
    """out[i] = src[idx[i]] (16-bit elements, one launch); `compact`: idx is gather16_compact_map's (n / 8, 2) form"""
    n = out.numel()
    if src.dtype not in (torch.bfloat16, torch.float16) or out.dtype != src.dtype or idx.dtype != torch.int32 or n % 8 or \
            not (src.is_contiguous() and out.is_contiguous() and idx.is_contiguous()) or idx.numel() != (n // 4 if compact else n) or \
            (out.data_ptr() | idx.data_ptr()) & 15:
        raise RuntimeError("gather16: 16-bit src / out, int32 map, whole 16-byte groups")
    p = L.Gather16Args()
    p.count, p.mode = n, 1 if compact else 0
    p.src, p.map, p.out = src.data_ptr(), idx.data_ptr(), out.data_ptr()
    p.stream = L.stream_handle(out)
    lib.check(lib.dll.segm_gather16(p), "gather16")
    return out


def transpose_add(lib: L.SegmLib, x: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B, R, C) contiguous -> (B, C, R) contiguous (+ add, which already has the output layout)."""
    if x.dim() != 3 or not x.is_contiguous():
        raise RuntimeError("transpose_add: x must be a contiguous (B, R, C) tensor")
    B, R, Cc = x.shape
    out = torch.empty(B, Cc, R, dtype=x.dtype, device=x.device)
    if add is not None and (add.shape != out.shape or add.dtype != x.dtype or not add.is_contiguous()):
        raise RuntimeError("transpose_add: add must be a contiguous (B, C, R) tensor of x's dtype")
    a = L.TransposeArgs()
    a.batch, a.rows, a.cols, a.dtype = B, R, Cc, L.dtype_code(x)
    a.in_, a.add, a.out = x.data_ptr(), (add.data_ptr() if add is not None else None), out.data_ptr()
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_transpose_add(a), "transpose_add")
    return out


# ---------------------------------------------------------------------------------------------------------
# volume -> tokens with LayerNorm
# ---------------------------------------------------------------------------------------------------------
def layernorm_tokens_supported(x: torch.Tensor) -> bool:
    if x.dim() != 3 or not x.is_contiguous() or x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return False
    n = 4 if x.dtype == torch.float32 else 8
    return x.shape[1] % n == 0 and x.shape[2] % n == 0 and x.shape[1] <= (192 if x.dtype == torch.float32 else 384)


def _ln_args(x, gamma, eps):
    B, Cc, S = x.shape
    a = L.LayerNormArgs()
    a.batch, a.channels, a.dtype, a.spatial, a.eps = B, Cc, L.dtype_code(x), S, float(eps)
    a.x, a.gamma, a.stream = x.data_ptr(), gamma.data_ptr(), L.stream_handle(x)
    return a


def layernorm_tokens_fwd(lib: L.SegmLib, x, gamma, beta, eps=1e-5):
    """x (B, C, S) contiguous -> (y (B, S, C), mean (B, S), rstd (B, S)); gamma / beta fp32 (C)."""
    if not layernorm_tokens_supported(x):
        raise RuntimeError("layernorm_tokens: unsupported shape / dtype / layout")
    B, Cc, S = x.shape
    gamma, beta = gamma.float().contiguous(), beta.float().contiguous()
    y = torch.empty(B, S, Cc, dtype=x.dtype, device=x.device)
    mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, S, dtype=torch.float32, device=x.device)
    a = _ln_args(x, gamma, eps)
    a.y, a.beta, a.mean, a.rstd = y.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    lib.check(lib.dll.segm_layernorm_tokens_fwd(a), "layernorm_tokens_fwd")
    return y, mean, rstd


def layernorm_tokens_bwd(lib: L.SegmLib, x, dy, mean, rstd, gamma):
    """-> (dx (B, C, S), dgamma (C) fp32, dbeta (C) fp32)."""
    B, Cc, S = x.shape
    if dy.shape != (B, S, Cc) or dy.dtype != x.dtype:
        raise RuntimeError("layernorm_tokens: dy must be (B, S, C) of x's dtype")
    dy = dy.contiguous()
    gamma = gamma.float().contiguous()
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws_bytes = lib.dll.segm_layernorm_tokens_workspace_bytes(B, Cc, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    a = _ln_args(x, gamma, 0.0)
    a.mean, a.rstd, a.dy, a.dx = mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr()
    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    lib.check(lib.dll.segm_layernorm_tokens_bwd(a), "layernorm_tokens_bwd")
    return dx, dgamma, dbeta


# ---------------------------------------------------------------------------------------------------------
# training-step glue: gradient clipping + SGD over a tensor list, cross entropy with its gradient
# ---------------------------------------------------------------------------------------------------------
def sgd_clip_step(lib: L.SegmLib, params, grads, momenta, lr: float, momentum: float, weight_decay: float,
                  nesterov: bool, max_norm: float, loss_scale: Optional[torch.Tensor] = None,
                  found_inf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In place: clip_grad_norm_(max_norm) + SGD step over lists of contiguous fp32 tensors (params / momenta updated,
    grads only read).  Returns the 4-float workspace head {update coefficient, gradient norm, skipped, -} (device tensor).
    loss_scale (1-element fp32 device tensor): the gradients carry that factor (GradScaler) - the norm and the clip are those of
    the unscaled gradients, and a non-finite norm skips the step and writes 1 to `found_inf` (1-element fp32), else 0."""
    n = len(params)
    if not (len(grads) == len(momenta) == n):
        raise RuntimeError("sgd_clip_step: params, grads and momenta must have the same length")
    if n == 0:
        return torch.ones(4)
    for p, g, m in zip(params, grads, momenta):
        if not (p.dtype == g.dtype == m.dtype == torch.float32) or not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous()) \
                or not (p.numel() == g.numel() == m.numel()) or not (p.device == g.device == m.device == params[0].device):
            raise RuntimeError("sgd_clip_step: tensors must be contiguous fp32, same size per entry, one device")
    ptr_t, i64_t = C.c_void_p * n, C.c_int64 * n
    pp, gp, mp = ptr_t(*[p.data_ptr() for p in params]), ptr_t(*[g.data_ptr() for g in grads]), ptr_t(*[m.data_ptr() for m in momenta])
    ne = i64_t(*[p.numel() for p in params])
    ws_bytes = lib.dll.segm_sgd_clip_step_workspace_bytes(n, C.cast(ne, C.c_void_p))
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=params[0].device)
    a = L.SgdArgs()
    a.ntensors, a.nesterov = n, int(bool(nesterov))
    a.params, a.grads, a.momenta, a.numel = C.cast(pp, C.c_void_p), C.cast(gp, C.c_void_p), C.cast(mp, C.c_void_p), C.cast(ne, C.c_void_p)
    a.lr, a.momentum, a.weight_decay, a.max_norm = lr, momentum, weight_decay, max_norm
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    a.stream = L.stream_handle(params[0])
    for t in (loss_scale, found_inf):
        if t is not None and (t.dtype != torch.float32 or t.numel() != 1 or t.device != params[0].device):
            raise RuntimeError("sgd_clip_step: loss_scale / found_inf must be 1-element fp32 tensors on the parameters' device")
    a.loss_scale = loss_scale.data_ptr() if loss_scale is not None else None
    a.found_inf = found_inf.data_ptr() if found_inf is not None else None
    lib.check(lib.dll.segm_sgd_clip_step(a), "sgd_clip_step")
    return ws[:4]


def cross_entropy_supported(logits: torch.Tensor, labels: torch.Tensor) -> bool:
    return logits.dim() >= 2 and 1 <= logits.shape[1] <= 16 and labels.dtype == torch.int64 and \
        logits.dtype in (torch.float32, torch.float16, torch.bfloat16) and \
        tuple(labels.shape) == (logits.shape[0],) + tuple(logits.shape[2:]) and logits.numel() > 0


def cross_entropy(lib: L.SegmLib, logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100):
    """-> (loss_sum fp32 scalar, count fp32 scalar, dlogits): sum over counted voxels of the cross entropy, the number of
    counted voxels, and d(loss_sum)/d(logits) (logits' dtype and shape)."""
    if not cross_entropy_supported(logits, labels):
        raise RuntimeError("cross_entropy: logits (B, C <= 16, *spatial) fp32 / fp16 / bf16 and int64 labels (B, *spatial)")
    logits, labels = logits.contiguous(), labels.contiguous()
    B, Cc = logits.shape[:2]
    S = logits.numel() // (B * Cc)
    nparts = lib.dll.segm_cross_entropy_partials(B, S)
    parts = torch.empty(2, nparts, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    a = L.CrossEntropyArgs()
    a.batch, a.classes, a.dtype = B, Cc, L.dtype_code(logits)
    a.spatial, a.ignore_index = S, ignore_index
    a.logits, a.labels, a.dlogits = logits.data_ptr(), labels.data_ptr(), dlogits.data_ptr()
    a.loss_partial, a.count_partial = parts[0].data_ptr(), parts[1].data_ptr()
    a.stream = L.stream_handle(logits)
    lib.check(lib.dll.segm_cross_entropy(a), "cross_entropy")
    tot = parts.sum(1)
    return tot[0], tot[1], dlogits


# ---------------------------------------------------------------------------------------------------------
# single-token decode steps
# ---------------------------------------------------------------------------------------------------------
def conv1d_update(lib: L.SegmLib, x, conv_state, weight, bias=None, silu=False):
    """x (B, D), conv_state (B, D, W) updated in place, weight (D, W) / bias (D) fp32 -> out (B, D) of x's dtype."""
    if x.dim() != 2 or conv_state.dim() != 3 or tuple(conv_state.shape[:2]) != tuple(x.shape) or conv_state.dtype != x.dtype:
        raise RuntimeError("conv1d_update: x (B, D) and conv_state (B, D, W) of one dtype")
    B, D = x.shape
    W = conv_state.shape[2]
    if tuple(weight.shape) != (D, W) or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise RuntimeError("conv1d_update: weight must be a contiguous fp32 (D, W) tensor")
    if bias is not None and (tuple(bias.shape) != (D,) or bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise RuntimeError("conv1d_update: bias must be a contiguous fp32 (D) tensor")
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    a = L.Conv1dUpdateArgs()
    a.batch, a.dim, a.width, a.silu, a.dtype = B, D, W, int(bool(silu)), L.dtype_code(x)
    a.x, a.x_stride_b, a.x_stride_d = x.data_ptr(), x.stride(0), x.stride(1)
    a.conv_state = conv_state.data_ptr()
    a.state_stride_b, a.state_stride_d, a.state_stride_w = conv_state.stride()
    a.out, a.out_stride_b, a.out_stride_d = out.data_ptr(), out.stride(0), out.stride(1)
    a.weight, a.bias, a.stream = weight.data_ptr(), L.fptr(bias), L.stream_handle(x)
    lib.check(lib.dll.segm_causal_conv1d_update(a), "causal_conv1d_update")
    return out


def state_update(lib: L.SegmLib, state, x, dt, A, Bm, Cm, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state (B, D, N) updated in place; x, dt, z (B, D); Bm, Cm (B, N); A (D, N), D, dt_bias (D) fp32 -> out (B, D)."""
    if state.dim() != 3 or x.dim() != 2 or tuple(state.shape[:2]) != tuple(x.shape):
        raise RuntimeError("state_update: state (B, D, N), x (B, D)")
    B, Dm, N = state.shape
    for name, t, shape in (("dt", dt, (B, Dm)), ("B", Bm, (B, N)), ("C", Cm, (B, N))) + ((("z", z, (B, Dm)),) if z is not None else ()):
        if tuple(t.shape) != shape or t.dtype != x.dtype:
            raise RuntimeError(f"state_update: {name} must be {shape} of x's dtype")
    for name, t, shape in (("A", A, (Dm, N)),) + ((("D", D, (Dm,)),) if D is not None else ()) + \
            ((("dt_bias", dt_bias, (Dm,)),) if dt_bias is not None else ()):
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"state_update: {name} must be a contiguous fp32 {shape} tensor")
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    a = L.StateUpdateArgs()
    a.batch, a.dim, a.dstate, a.dt_softplus = B, Dm, N, int(bool(dt_softplus))
    a.dtype, a.state_dtype = L.dtype_code(x), L.dtype_code(state)
    a.state = state.data_ptr()
    a.state_stride_b, a.state_stride_d, a.state_stride_n = state.stride()
    for name, t in (("x", x), ("dt", dt), ("out", out)):
        setattr(a, name, t.data_ptr()); setattr(a, name + "_stride_b", t.stride(0)); setattr(a, name + "_stride_d", t.stride(1))
    if z is not None:
        a.z, a.z_stride_b, a.z_stride_d = z.data_ptr(), z.stride(0), z.stride(1)
    a.B, a.B_stride_b, a.B_stride_n = Bm.data_ptr(), Bm.stride(0), Bm.stride(1)
    a.C, a.C_stride_b, a.C_stride_n = Cm.data_ptr(), Cm.stride(0), Cm.stride(1)
    a.A, a.D, a.dt_bias, a.stream = A.data_ptr(), L.fptr(D), L.fptr(dt_bias), L.stream_handle(x)
    lib.check(lib.dll.segm_selective_state_update(a), "selective_state_update")
    return out


# ---------------------------------------------------------------------------------------------------------
# row-streaming projection (tall activations)
# ---------------------------------------------------------------------------------------------------------
def linear_rows_supported(x2: torch.Tensor, w: torch.Tensor, n_out: Optional[int] = None) -> bool:
    """x2 (rows, K) view with unit column stride, w (N, K)."""
    if x2.dim() != 2 or w.dim() != 2 or x2.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x2.dtype:
        return False
    K, N = x2.shape[1], w.shape[0]
    return w.shape[1] == K and K % 8 == 0 and K <= 2048 and N % 4 == 0 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and \
        x2.data_ptr() % 16 == 0 and x2.shape[0] > 0


def linear_rows(lib: L.SegmLib, x2: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """y (rows, N) = x2 (rows, K) @ w (N, K)^T + bias; `out` may be a column slice of a wider row-major tensor;
    `accumulate` adds to what `out` holds."""
    if not linear_rows_supported(x2, w):
        raise RuntimeError("linear_rows: x (rows, K <= 2048, K % 8 == 0) with unit column stride and 16-byte rows, w (N % 4 == 0, K), bf16 / fp16")
    M, K = x2.shape
    N = w.shape[0]
    w = w.contiguous()
    y = torch.empty(M, N, dtype=x2.dtype, device=x2.device) if out is None else out
    if tuple(y.shape) != (M, N) or y.dtype != x2.dtype or y.stride(1) != 1 or y.stride(0) % 4 or y.data_ptr() % 8:
        raise RuntimeError("linear_rows: `out` must be (rows, N) of x's dtype, unit column stride, 8-byte rows")
    if bias is not None:
        bias = bias.float().contiguous()
    a = L.LinearArgs()
    if accumulate and out is None:
        raise RuntimeError("linear_rows: accumulate needs `out`")
    a.rows, a.k, a.n, a.dtype, a.accumulate = M, K, N, L.dtype_code(x2), int(bool(accumulate))
    a.x, a.x_stride_row = x2.data_ptr(), x2.stride(0)
    a.w, a.bias = w.data_ptr(), L.fptr(bias)
    a.y, a.y_stride_row = y.data_ptr(), y.stride(0)
    a.stream = L.stream_handle(x2)
    lib.check(lib.dll.segm_linear_rows(a), "linear_rows")
    return y


def skinny_tn_supported(wide: torch.Tensor, skinny: torch.Tensor) -> bool:
    return bool(wide.dim() == 2 and skinny.dim() == 2 and wide.shape[0] == skinny.shape[0] and skinny.shape[1] <= 32 and
                wide.dtype in (torch.bfloat16, torch.float16) and skinny.dtype == wide.dtype and wide.stride(1) == 1 and
                skinny.stride(1) == 1 and wide.shape[0] > 0 and
                # whole rows as 16-byte loads: eight channels per thread, at most 256 threads per row
                wide.shape[1] % 8 == 0 and wide.shape[1] <= 2048 and wide.stride(0) % 8 == 0 and wide.data_ptr() % 16 == 0)


def skinny_tn(lib: L.SegmLib, wide: torch.Tensor, skinny: torch.Tensor) -> torch.Tensor:
    """wide (k, m)^T @ skinny (k, n <= 32) -> (m, n) fp32: the dt_proj weight gradient as a streaming reduction"""
    if not skinny_tn_supported(wide, skinny):
        raise RuntimeError("skinny_tn: wide (k, m % 8 == 0, m <= 2048, 16-byte aligned rows), skinny (k, n <= 32), one 16-bit dtype, unit column strides")
    K, M = wide.shape
    N = skinny.shape[1]
    out = torch.empty(M, N, dtype=torch.float32, device=wide.device)
    ws_bytes = lib.dll.segm_skinny_tn_workspace_bytes(M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=wide.device)
    a = L.SkinnyTnArgs()
    a.k, a.m, a.n, a.dtype = K, M, N, L.dtype_code(wide)
    a.wide, a.wide_stride_row = wide.data_ptr(), wide.stride(0)
    a.skinny, a.skinny_stride_row = skinny.data_ptr(), skinny.stride(0)
    a.out, a.workspace, a.workspace_bytes = out.data_ptr(), ws.data_ptr(), ws_bytes
    a.stream = L.stream_handle(wide)
    lib.check(lib.dll.segm_skinny_tn(a), "skinny_tn")
    return out


def channel_sum_supported(x: torch.Tensor) -> bool:
    """(B, C, *spatial) whose voxels are one unit-stride run per (b, c) row (dense or padded channel stride)"""
    if x.dim() < 3 or x.dtype not in (torch.float32, torch.float16, torch.bfloat16) or x.numel() == 0:
        return False
    if x.shape[0] > 65535 or x.shape[1] > 65535:
        return False
    run = 1
    for size, stride in zip(reversed(x.shape[2:]), reversed(x.stride()[2:])):
        if size != 1 and stride != run:
            return False
        run *= size
    return x.stride(1) >= run or x.shape[1] == 1


def channel_sum(lib: L.SegmLib, x: torch.Tensor) -> torch.Tensor:
    """sum over the batch and the voxels of x (B, C, *spatial) -> (C,) fp32: a convolution's bias gradient"""
    if not channel_sum_supported(x):
        raise RuntimeError("channel_sum: (B, C, *spatial) with unit-stride voxels per (b, c) row, fp32 / fp16 / bf16")
    Bn, Cn = x.shape[0], x.shape[1]
    S = 1
    for n in x.shape[2:]:
        S *= n
    out = torch.empty(Cn, dtype=torch.float32, device=x.device)
    ws_bytes = lib.dll.segm_channel_sum_workspace_bytes(Bn, Cn, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    a = L.ChannelSumArgs()
    a.x, a.stride_batch, a.stride_channel, a.spatial = x.data_ptr(), x.stride(0), (x.stride(1) if Cn > 1 else S), S
    a.batch, a.channels, a.dtype = Bn, Cn, L.dtype_code(x)
    a.out, a.workspace, a.workspace_bytes = out.data_ptr(), ws.data_ptr(), ws_bytes
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_channel_sum(a), "channel_sum")
    return out


def pointwise_cf_supported(x3: torch.Tensor, cout: int) -> bool:
    """x3 (B, Cin <= 96, S) 16-bit with contiguous voxels, S % 64 == 0, 16-byte aligned channel rows; Cout <= 96"""
    return bool(x3.dim() == 3 and x3.dtype in (torch.bfloat16, torch.float16) and x3.shape[1] <= 96 and cout <= 96
                and x3.shape[2] % 64 == 0 and x3.stride(2) == 1 and x3.stride(0) % 8 == 0 and x3.stride(1) % 8 == 0
                and x3.data_ptr() % 16 == 0)


def pointwise_cf(lib: L.SegmLib, x3: torch.Tensor, w2: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """y (B, Cout, S) = w2 (Cout, Cin) @ x3[b] (Cin, S) + bias, channel-first in and out (a 1x1x1 convolution on NCDHW
    activations flattened over their voxels, or on channel slices of them).  `out` + `accumulate`: y += ... ."""
    B, Cin, S = x3.shape
    Cout = w2.shape[0]
    if not pointwise_cf_supported(x3, Cout) or w2.shape[1] != Cin or w2.dtype != x3.dtype:
        raise RuntimeError("pointwise_cf: x (B, Cin <= 96, S % 64 == 0) 16-bit with contiguous 16-byte aligned voxel rows, "
                           "w (Cout <= 96, Cin) of the same dtype")
    if Cin % 8:                                            # pad the weight rows to a multiple of 8 (zero columns)
        w2 = torch.nn.functional.pad(w2, (0, 8 - Cin % 8))
    w2 = w2.contiguous()
    y = volume_empty(B, Cout, (S,), x3.dtype, x3.device) if out is None else out
    if tuple(y.shape) != (B, Cout, S) or y.dtype != x3.dtype or y.stride(2) != 1 or y.stride(0) % 4 or y.stride(1) % 4 \
            or y.data_ptr() % 8:
        raise RuntimeError("pointwise_cf: `out` must be (B, Cout, S) of x's dtype with contiguous 8-byte aligned voxel rows")
    if accumulate and out is None:
        raise RuntimeError("pointwise_cf: accumulate needs `out`")
    if bias is not None:
        bias = bias.float().contiguous()
    a = L.PointwiseArgs()
    a.batch, a.cin, a.cout, a.dtype, a.accumulate, a.spatial = B, Cin, Cout, L.dtype_code(x3), int(bool(accumulate)), S
    a.x, a.x_stride_b, a.x_stride_c = x3.data_ptr(), x3.stride(0), x3.stride(1)
    a.w, a.w_stride, a.bias = w2.data_ptr(), w2.shape[1], L.fptr(bias)
    a.y, a.y_stride_b, a.y_stride_c = y.data_ptr(), y.stride(0), y.stride(1)
    a.stream = L.stream_handle(x3)
    lib.check(lib.dll.segm_pointwise_cf(a), "pointwise_cf")
    return y


def _stem_geometry(weight: torch.Tensor):
    """(kernel size, stride) the thin-input kernels run a weight of this shape with: 7^3 -> stride 2 (the stem), 3^3 -> stride 1"""
    k = tuple(weight.shape[2:])
    return (7, 2) if k == (7, 7, 7) else ((3, 1) if k == (3, 3, 3) else (0, 0))


def stem_conv_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x (B, Cin <= 4, D, H, W); weight (Cout <= 48, Cin, 7, 7, 7) (stride 2, padding 3: D, H even, W % 32 == 0) or
    (Cout <= 48, Cin, 3, 3, 3) (stride 1, padding 1: W % 16 == 0); a wave's 8 output tiles are TX along x (the largest of 8, 4, 2, 1
    dividing Wout / 16) times 8 / TX rows, which must divide Hout"""
    if not (x.dim() == 5 and weight.dim() == 5 and weight.shape[1] == x.shape[1] and x.shape[1] <= 4 and weight.shape[0] <= 48):
        return False
    k, s = _stem_geometry(weight)
    if not k or x.shape[2] % s or x.shape[3] % s or x.shape[4] % (16 * s):
        return False
    xt, hout = x.shape[4] // (16 * s), x.shape[3] // s
    return any(xt % tx == 0 and hout % (8 // tx) == 0 for tx in (8, 4, 2, 1))


def pack_stem_weight(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(Cout, Cin <= 4, k, k, k) -> (Cout, k, k, 8, 4): [co][kz][ky][kx slot][ci], slots >= k and missing channels zero"""
    w = weight.permute(0, 2, 3, 4, 1)                                  # co, kz, ky, kx, ci
    w = torch.nn.functional.pad(w, (0, 4 - w.shape[-1], 0, 8 - w.shape[-2]))
    return w.contiguous().to(dtype)


def stem_channel_last4(x: torch.Tensor) -> torch.Tensor:
    """(B, Cin <= 4, D, H, W) -> (B, D, H, W, 4) contiguous, missing channels zero: the input layout of the thin-input kernels"""
    return torch.nn.functional.pad(x.permute(0, 2, 3, 4, 1), (0, 4 - x.shape[1])).contiguous()


def stem_conv_fwd(lib: L.SegmLib, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  x4: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = conv3d(x, weight, bias, stride 2, padding 3) for the 7^3 stem, conv3d(x, weight, bias, stride 1, padding 1) for a 3^3
    weight (x, weight of one 16-bit dtype, at most 4 input channels) -> (B, Cout, D / s, H / s, W / s)"""
    if not stem_conv_supported(x, weight) or x.dtype not in (torch.bfloat16, torch.float16) or weight.dtype != x.dtype:
        raise RuntimeError("stem_conv_fwd: x (B, Cin <= 4, D, H, W) and weight (Cout <= 48, Cin, 7, 7, 7) [stride 2: D, H even, W % 32 == 0] "
                           "or (Cout <= 48, Cin, 3, 3, 3) [stride 1: W % 16 == 0] of one 16-bit dtype")
    B, Cin, D, H, W = x.shape
    k, s = _stem_geometry(weight)
    if x4 is None:
        x4 = stem_channel_last4(x)
    wp = pack_stem_weight(weight, x.dtype)
    y = volume_empty(B, weight.shape[0], (D // s, H // s, W // s), x.dtype, x.device)
    if bias is not None:
        bias = bias.float().contiguous()
    a = L.StemArgs()
    a.batch, a.cout, a.din, a.hin, a.win, a.dtype = B, weight.shape[0], D, H, W, L.dtype_code(x)
    a.x4, a.w_packed, a.bias, a.y = x4.data_ptr(), wp.data_ptr(), L.fptr(bias), y.data_ptr()
    a.kernel_size, a.stride, a.y_channel_stride = k, s, y.stride(1)
    a.stream = L.stream_handle(x)
    lib.check(lib.dll.segm_stem_conv_fwd(a), "stem_conv_fwd")
    return y


def stem_wgrad_supported(x4: torch.Tensor, cout: int, kernel_size: int = 7) -> bool:
    """x4 (B, D, H, W, 4); kernel 7 (stride 2): D, H even and W in {64, 128, 256}; kernel 3 (stride 1): W in {32, 64, 128}; cout <= 48"""
    if not (x4.dim() == 5 and x4.shape[4] == 4 and cout <= 48):
        return False
    if kernel_size == 7:
        return x4.shape[1] % 2 == 0 and x4.shape[2] % 2 == 0 and x4.shape[3] in (64, 128, 256)
    return kernel_size == 3 and x4.shape[3] in (32, 64, 128)


def stem_conv_wgrad(lib: L.SegmLib, x4: torch.Tensor, dy: torch.Tensor, cin: int, kernel_size: int = 7) -> torch.Tensor:
    """dW (Cout, cin, k, k, k) fp32 of conv3d(x, W, stride 2, padding 3) [k = 7] or conv3d(x, W, stride 1, padding 1) [k = 3] from the
    channel-last-4 input x4 (B, D, H, W, 4) and dy (B, Cout, D / s, H / s, W / s) (dense channels), one 16-bit dtype"""
    B, D, H, W, _ = x4.shape
    cout = dy.shape[1]
    s = 2 if kernel_size == 7 else 1
    if (not stem_wgrad_supported(x4, cout, kernel_size) or x4.dtype not in (torch.bfloat16, torch.float16) or dy.dtype != x4.dtype
            or tuple(dy.shape) != (B, cout, D // s, H // s, W // s) or not 1 <= cin <= 4):
        raise RuntimeError("stem_conv_wgrad: x4 (B, D, H, W, 4) and dy (B, Cout <= 48, D / s, H / s, W / s) of one 16-bit dtype; "
                           "kernel 7: D, H even, W in {64, 128, 256}; kernel 3: W in {32, 64, 128}")
    x4 = x4.contiguous()
    if not channel_dense(dy) or dy.stride(1) % 8:
        dy = dy.contiguous()
    cout16 = (cout + 15) // 16 * 16
    slots = 8 if kernel_size == 7 else 4
    dwp = torch.empty(kernel_size, kernel_size, cout16, slots, 4, dtype=torch.float32, device=x4.device)
    nbytes = lib.dll.segm_stem_conv_wgrad_workspace_bytes2(B, cout, D, H, kernel_size, s)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x4.device)
    a = L.StemWgradArgs()
    a.batch, a.cout, a.din, a.hin, a.win, a.dtype = B, cout, D, H, W, L.dtype_code(x4)
    a.x4, a.dy, a.dw_packed = x4.data_ptr(), dy.data_ptr(), dwp.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    a.kernel_size, a.stride, a.dy_channel_stride = kernel_size, s, dy.stride(1)
    a.stream = L.stream_handle(x4)
    lib.check(lib.dll.segm_stem_conv_wgrad(a), "stem_conv_wgrad")
    # [kz][ky][co][kx slot][ci] -> (co, ci, kz, ky, kx)
    return dwp[:, :, :cout, :kernel_size, :cin].permute(2, 4, 0, 1, 3).contiguous()


WGEMM_TN, WGEMM_NT = 0, 1


def wgrad_gemm_tn_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    """a (K, M), b (K, N): 16-bit, same dtype, unit column stride, M, N <= 1024"""
    return bool(a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.dtype == b.dtype
                and a.dtype in (torch.bfloat16, torch.float16) and a.stride(1) == 1 and b.stride(1) == 1
                and a.shape[1] <= 1024 and b.shape[1] <= 1024 and a.stride(0) >= a.shape[1] and b.stride(0) >= b.shape[1])


def wgrad_gemm_nt_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    """a (B, M, K), b (B, N, K): 16-bit, same dtype, unit stride along K, K % 32 == 0, M, N <= 96, 16-byte aligned rows"""
    return bool(a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2] and a.dtype == b.dtype
                and a.dtype in (torch.bfloat16, torch.float16) and a.stride(2) == 1 and b.stride(2) == 1 and a.shape[2] % 32 == 0
                and a.shape[1] <= 96 and b.shape[1] <= 96 and not any(t.stride(i) % 8 for t in (a, b) for i in (0, 1))
                and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def wgrad_gemm(lib: L.SegmLib, a: torch.Tensor, b: torch.Tensor, layout: int) -> torch.Tensor:
    """fp32 (M, N): a^T b for a (K, M), b (K, N) (WGEMM_TN), or sum_i a[i] b[i]^T for a (B, M, K), b (B, N, K) (WGEMM_NT)"""
    if layout == WGEMM_TN:
        if not wgrad_gemm_tn_supported(a, b):
            raise RuntimeError("wgrad_gemm TN: a (K, M), b (K, N) 16-bit with unit column stride, M, N <= 1024")
        K, M = a.shape
        N, batch = b.shape[1], 1
        strides = (a.stride(0), 0, b.stride(0), 0)
    else:
        if not wgrad_gemm_nt_supported(a, b):
            raise RuntimeError("wgrad_gemm NT: a (B, M, K), b (B, N, K) 16-bit, unit stride along K % 32 == 0, M, N <= 96, 16-byte aligned rows")
        batch, M, K = a.shape
        N = b.shape[1]
        strides = (a.stride(1), a.stride(0), b.stride(1), b.stride(0))
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    nbytes = lib.dll.segm_wgrad_gemm_workspace_bytes(layout, M, N, K, batch)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    g = L.WgradGemmArgs()
    g.layout, g.dtype, g.m, g.n, g.k, g.batch = layout, L.dtype_code(a), M, N, K, batch
    g.a, g.a_stride_row, g.a_stride_batch = a.data_ptr(), strides[0], strides[1]
    g.b, g.b_stride_row, g.b_stride_batch = b.data_ptr(), strides[2], strides[3]
    g.out, g.workspace, g.workspace_bytes, g.stream = out.data_ptr(), ws.data_ptr(), nbytes, L.stream_handle(a)
    lib.check(lib.dll.segm_wgrad_gemm(g), "wgrad_gemm")
    return out


# ---------------------------------------------------------------------------------------------------------
# device guard
# ---------------------------------------------------------------------------------------------------------
# The reference's native ops run under a CUDAGuard on their first tensor's device (selective_scan.cpp:326-327,
# causal_conv1d.cpp:178-179): a kernel launch goes to the CURRENT device, whatever device the pointers and the stream
# belong to.  Every launching wrapper above therefore runs with the device of its first CUDA tensor current (a model on
# cuda:1 while the process's current device is cuda:0 must work); nothing happens - one integer compare - when it already is.
def _first_cuda_tensor(args):
    """the first CUDA tensor among the arguments, looking into lists / tuples and into the keyword dicts of the `*_multi` calls"""
    for a in args:
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a
        elif isinstance(a, dict):
            t = _first_cuda_tensor(a.values())
            if t is not None:
                return t
        elif isinstance(a, (list, tuple)) and a:
            t = _first_cuda_tensor(a)
            if t is not None:
                return t
    return None


def _device_guard(fn):
    import functools

    @functools.wraps(fn)
    def guarded(lib, *args, **kw):
        t = _first_cuda_tensor(args)
        if t is None:
            t = _first_cuda_tensor(kw.values())
        if t is not None and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(lib, *args, **kw)
        return fn(lib, *args, **kw)
    return guarded


for _name in ("scan_fwd", "scan_bwd", "conv1d_fwd", "conv1d_bwd", "conv3d_k3_wgrad", "conv3d_k3_fwd", "instnorm_fwd",
              "instnorm_bwd", "transpose_add", "layernorm_tokens_fwd", "layernorm_tokens_bwd", "sgd_clip_step", "cross_entropy",
              "conv1d_update", "state_update", "linear_rows", "skinny_tn", "pointwise_cf", "stem_conv_fwd", "stem_conv_wgrad", "wgrad_gemm",
              "scan_fwd_multi", "scan_bwd_multi", "conv1d_fwd_multi", "conv1d_bwd_multi", "channel_sum", "depth_to_space2",
              "space_to_depth2"):
    globals()[_name] = _device_guard(globals()[_name])
