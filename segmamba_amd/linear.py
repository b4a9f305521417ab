"""Projection GEMMs of the hot path (1x1x1 convolutions, patch convolutions, Mamba in/out/x/dt projections).

Tall 16-bit activations with at most 192 channels (the Mamba projections of stages 0 / 1) go through the library's
row-streaming MFMA kernel (csrc/linear.hip: 1.0 - 2.0x the BLAS call on every such shape, profiles/r02_linear.log);
everything else stays on the BLAS library (rocBLAS / hipBLASLt through torch - plain library GEMMs), where two things
about how they are *called* matter on MI355X (profiles/r01_bench_step_kernels_v5.txt, tools/gpu_torch_prof.py):

  * weight gradients are  dW = dY^T X  with a reduction over K = B*D*H*W rows (up to 4.2 M) and only 3 .. 192 output
    rows / columns.  The library runs such a shape as ONE small tile looping over all of K (1.3 - 5 ms each, ~40 ms
    of a 283 ms training step).  `tn_matmul` cuts K into slabs that become the batch dimension of a batched GEMM
    (thousands of independent tiles) and adds the partial products - "split-K", done above the library.
  * a GEMM can read either operand transposed for free, so a 1x1x1 convolution can consume channel-first or
    channel-last activations as they are and always emit the channel-first tensor the next operator wants; no
    transposing copy is materialised around it.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

_SLAB = 4096          # rows of K per partial product
_MIN_K = 32768        # below this one GEMM is fine
_FORCE_SPLIT = False  # tests: exercise the split path on CPU tensors too
# The library's row-streaming projection kernel (csrc/linear.hip) for tall activations: 3.4 - 4.9 TB/s against 1.4 - 2.9 TB/s
# for the BLAS calls at stage 0, 1.7 - 2.4 against 1.2 - 2.3 at stage 1 (profiles/r02_linear.log).  SEGM_LINEAR_HIP=0 -> BLAS.
_ROWS_HIP = os.environ.get("SEGM_LINEAR_HIP", "1") == "1"
_ROWS_MAX_K = int(os.environ.get("SEGM_LINEAR_ROWS_MAX_K", "192"))  # widest contraction routed to segm_linear_rows: its streamed-W form for K > 192
                                                                    # (round 6) is behind the vendor GEMM - at the stage-2 / 3 row counts AND at config 1's
                                                                    # 524 288 rows, where routing it by mistake cost 3.6 ms of the block's 11.5 ms forward
_ROWS_MIN = int(os.environ.get("SEGM_LINEAR_ROWS_MIN", "32768"))     # rows below which the BLAS call stays (round 6 measured the
# library kernels on the stage-2 / 3 shapes: 6 - 25 us against 5.6 - 12.7 us for the vendor GEMM, step +0.2 ms: the threshold stays)
# Channel-first 1x1x1 convolutions (csrc/pointwise.hip): the BLAS route runs y[b] = W x[b] on strided views at ~1.6 TB/s and
# adds the bias in a further pass; SEGM_POINTWISE_HIP=0 restores it.
_PW_HIP = os.environ.get("SEGM_POINTWISE_HIP", "1") == "1"
_PW_MIN = 32768       # voxels per channel below which the BLAS call stays
# The weight gradients of the 1x1x1 convolutions on channel-first volumes through csrc/wgrad_gemm.hip (layout NT: 135 - 270 us
# against 205 - 310 us for slab-batched BLAS GEMMs + a sum at 128^3, profiles/r02_wgrad_gemm_time.log); SEGM_WGRAD_GEMM_HIP=0
# restores those.  The token-major layout (TN: the Mamba projections): rounds 2 - 5 left it on the BLAS slabs (3 - 3.9 TB/s against
# 1.1 - 2.6 for the library's kernel); round 6's kernel (two chunks in flight per wave, 48-column blocks for n <= 48, whole 16-byte
# pieces on x_dbl's padded rows) is level or ahead on every shape the step has from 4096 tokens up - in_proj s0 59 vs 66 us, the
# stage-1 shapes 22 - 46 vs 34 - 58 us, the stage-2 shapes (ONE vendor GEMM of 52 us each before: no split over K) 21 - 35 us - and
# returns fp32 sums where the short-K vendor path rounded to 16 bits (profiles/r06_wgemm_tn_v3.log, _v4.log): default on, step
# -0.17 ms (r06_wgemm_tn_step2.log); SEGM_WGRAD_GEMM_TN=0 restores the vendor route.
_WG_HIP = os.environ.get("SEGM_WGRAD_GEMM_HIP", "1") == "1"
_WG_TN = os.environ.get("SEGM_WGRAD_GEMM_TN", "1") == "1"      # round 6: default on (two chunks in flight, 48-column blocks, padded rows)
_TN_MIN_K = int(os.environ.get("SEGM_WGRAD_GEMM_TN_MIN_K", "4096"))   # the TN kernel splits K over waves: also ahead of ONE GEMM on short K


def _on_device(t: torch.Tensor) -> bool:
    from . import lib as L
    return L.on_device(t)


def _rows_hip(x2: torch.Tensor, w: torch.Tensor, b):
    """x2 (rows, K) @ w (N, K)^T + b through segm_linear_rows, or None when the shape / layout is not the kernel's."""
    if not (_ROWS_HIP and _on_device(x2) and x2.shape[0] >= _ROWS_MIN and w.shape[1] <= _ROWS_MAX_K):
        return None
    from . import lib as L, ops_raw
    if not ops_raw.linear_rows_supported(x2, w):
        return None
    return ops_raw.linear_rows(L.get_lib(), x2, w, b)


def _split(K: int) -> int:
    """number of K slabs (a divisor of K close to K / _SLAB), 1 = do not split."""
    if K < _MIN_K:
        return 1
    s = K // _SLAB
    while s > 1 and K % s:
        s -= 1
    return s if K // max(s, 1) <= 4 * _SLAB else 1


_SKINNY = os.environ.get("SEGM_SKINNY_TN", "1") == "1"


_CHSUM = os.environ.get("SEGM_CHANNEL_SUM_HIP", "1") == "1"


def bias_grad(dy: torch.Tensor) -> torch.Tensor:
    """sum of dy (B, C, *spatial) over the batch and the voxels, fp32: a convolution's bias gradient.  The library's streaming
    reduction where the rows are unit-stride runs on the device (ATen's generic reduction reads these tensors at 0.3 - 0.7 TB/s),
    `dy.sum` otherwise."""
    if _CHSUM and _on_device(dy):
        from . import lib as L, ops_raw
        if ops_raw.channel_sum_supported(dy):
            return ops_raw.channel_sum(L.get_lib(), dy)
    return dy.sum([0] + list(range(2, dy.dim())), dtype=torch.float32)


def tn_matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T b for tall operands a (K, M), b (K, N), K >> M, N; fp32 result.  Row slices / column slices of larger
    matrices are fine (only views are taken)."""
    K = a.shape[0]
    if _SKINNY and K >= 4096 and _on_device(a) and b.shape[1] <= 32 < a.shape[1]:
        from . import lib as L, ops_raw
        if ops_raw.skinny_tn_supported(a, b):               # a tall-and-wide, b tall-and-skinny (dt_proj's weight gradient)
            return ops_raw.skinny_tn(L.get_lib(), a, b)
    if _WG_HIP and _WG_TN and K >= _TN_MIN_K and _on_device(a):
        from . import lib as L, ops_raw
        if ops_raw.wgrad_gemm_tn_supported(a, b):
            return ops_raw.wgrad_gemm(L.get_lib(), a, b, ops_raw.WGEMM_TN)
    s = _split(K) if (_on_device(a) or _FORCE_SPLIT) else 1
    if s == 1:
        if a.is_cuda:                                      # fp32 out of the GEMM itself: no cast launch, no 16-bit rounding of the sums
            try:
                return torch.mm(a.t(), b, out_dtype=torch.float32)
            except (TypeError, RuntimeError, NotImplementedError):
                pass
        return (a.t() @ b).float()
    part = torch.bmm(a.unflatten(0, (s, K // s)).transpose(1, 2), b.unflatten(0, (s, K // s)))      # (s, M, N)
    return part.sum(0, dtype=torch.float32)


def nt_matmul_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sum over the batch of a[i] b[i]^T for wide operands a (B, M, K), b (B, N, K) with unit stride along K; fp32."""
    B, M, K = a.shape
    if _WG_HIP and K >= _MIN_K and _on_device(a):
        from . import lib as L, ops_raw
        if ops_raw.wgrad_gemm_nt_supported(a, b):
            return ops_raw.wgrad_gemm(L.get_lib(), a, b, ops_raw.WGEMM_NT)
    s = _split(K) if (_on_device(a) or _FORCE_SPLIT) else 1
    if s == 1 or a.stride(2) != 1 or b.stride(2) != 1:
        return torch.matmul(a, b.transpose(1, 2)).sum(0, dtype=torch.float32)
    acc = None
    for i in range(B):
        pa = a[i].unflatten(1, (s, K // s)).transpose(0, 1)             # (s, M, K/s), rows strided
        pb = b[i].unflatten(1, (s, K // s)).transpose(0, 1)             # (s, N, K/s)
        p = torch.bmm(pa, pb.transpose(1, 2)).sum(0, dtype=torch.float32)
        acc = p if acc is None else acc + p
    return acc


def _masters(ctx, x, w, b):
    """the weight / bias of a Function in x's dtype (param_bank.low_precision: the step's 16-bit copy, no autograd edge); the
    Function returns their gradients in the dtype they came in (fp32 masters: straight from the fp32 accumulators)"""
    from .param_bank import low_precision
    ctx.w_dtype, ctx.b_dtype = w.dtype, (b.dtype if b is not None else None)
    ctx.has_bias = b is not None
    return low_precision(w, x.dtype), low_precision(b, x.dtype)


class _LinearCL(torch.autograd.Function):
    """y = x W^T + b on a channel-last x (..., Cin) already in the compute dtype; w, b in any dtype (fp32 masters under autocast)."""

    @staticmethod
    def forward(ctx, x, w, b):
        w, b = _masters(ctx, x, w, b)
        ctx.save_for_backward(x, w)
        y = _rows_hip(x.reshape(-1, x.shape[-1]), w, b)
        return F.linear(x, w, b) if y is None else y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.needs_input_grad[0]:
            from .param_bank import packed
            dx = _rows_hip(dy2, packed(w, "linear_t", lambda t: t.t().contiguous()), None)
            dx = (dy2 @ w if dx is None else dx).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            dw = tn_matmul(dy2, x.reshape(-1, x.shape[-1])).to(ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32).to(ctx.b_dtype)
        return dx, dw, db


def _pw_hip(w: torch.Tensor, x: torch.Tensor, b, into=None):
    """w (M, K) times every x[b] (K, S) (+ bias) through segm_pointwise_cf, or None when the shape / layout is not the kernel's;
    `into`: an existing (B, M, S) result the product is added to"""
    if not (_PW_HIP and _on_device(x) and x.shape[2] >= _PW_MIN and x.stride(2) == 1 and w.shape[1] <= 96):
        return None
    from . import lib as L, ops_raw
    if not ops_raw.pointwise_cf_supported(x, w.shape[0]) or w.dtype != x.dtype:
        return None
    return ops_raw.pointwise_cf(L.get_lib(), x, w, b, out=into, accumulate=into is not None)


def _bmm_w(w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """w (M, K) times every x[b] (K, S) -> (B, M, S) contiguous.  (torch.matmul would fold the batch into the rows of
    a transposed product, which costs a transposing copy of x and returns a channel-last result.)"""
    return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)


class _Pointwise(torch.autograd.Function):
    """1x1x1 convolution y[b] = W x[b] on x (B, Cin, S) in either memory order -> y (B, Cout, S) channel-first."""

    @staticmethod
    def forward(ctx, x, w, b):
        w, b = _masters(ctx, x, w, b)
        ctx.save_for_backward(x, w)
        y = _pw_hip(w, x, b)                                # bias in the kernel's accumulator initialisation
        if y is None:
            y = _bmm_w(w, x)
            if b is not None:
                y += b.view(1, -1, 1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        if dy.stride(2) != 1 and dy.stride(1) != 1:
            dy = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = _pw_hip(w.t(), dy, None)
            if dx is None:
                dx = _bmm_w(w.t(), dy)
        if ctx.needs_input_grad[1]:
            if x.stride(2) == 1 and dy.stride(2) == 1:
                dw = nt_matmul_rows(dy, x).to(ctx.w_dtype)
            else:                                                          # channel-last operands: (B*S, C) matrices
                xs, dys = x.transpose(1, 2), dy.transpose(1, 2)
                if x.stride(1) != 1:
                    xs = xs.contiguous()
                if dy.stride(1) != 1:
                    dys = dys.contiguous()
                dw = sum(tn_matmul(dys[i], xs[i]) for i in range(x.shape[0])).to(ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = bias_grad(dy).to(ctx.b_dtype)
        return dx, dw, db


class _PointwiseCat(torch.autograd.Function):
    """_Pointwise on the channel concatenation of xs without materialising it and without the adds: y = W[:, :c0] x0 + b, every
    later part added in place by the kernel (`accumulate`); one node, the weight gradient assembled by one cat."""

    @staticmethod
    def forward(ctx, w, b, *xs):
        w, b = _masters(ctx, xs[0], w, b)
        ctx.save_for_backward(w, *xs)
        out, c0 = None, 0
        for x in xs:
            wi = w[:, c0:c0 + x.shape[1]]
            c0 += x.shape[1]
            if out is None:
                out = _pw_hip(wi, x, b)
                if out is None:
                    out = _bmm_w(wi, x)
                    if b is not None:
                        out += b.view(1, -1, 1)
            else:
                y = _pw_hip(wi, x, None, into=out)
                out = y if y is not None else out + _bmm_w(wi, x)
        return out

    @staticmethod
    def backward(ctx, dy):
        w, *xs = ctx.saved_tensors
        if dy.stride(2) != 1:
            dy = dy.contiguous()
        dxs, dws, c0 = [], [], 0
        for i, x in enumerate(xs):
            wi = w[:, c0:c0 + x.shape[1]]
            c0 += x.shape[1]
            dx = None
            if ctx.needs_input_grad[2 + i]:
                dx = _pw_hip(wi.t(), dy, None)
                if dx is None:
                    dx = _bmm_w(wi.t(), dy)
            dxs.append(dx)
            if ctx.needs_input_grad[0]:
                dws.append(nt_matmul_rows(dy, x))
        dw = torch.cat(dws, dim=1).to(ctx.w_dtype) if ctx.needs_input_grad[0] else None
        db = bias_grad(dy).to(ctx.b_dtype) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return (dw, db, *dxs)


def pointwise_cat(xs, weight2d: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """pointwise(torch.cat(xs, 1), weight2d, bias) for channel-first parts (B, Ci, *spatial) -> (B, Cout, *spatial), or None when
    a part is not channel-first with contiguous voxels (the caller then sums per-part results)"""
    dt = _compute_dtype(xs[0])[0].dtype
    flat = [x.to(dt).flatten(2) for x in xs]
    if any(f.stride(2) != 1 for f in flat):
        return None
    y = _PointwiseCat.apply(weight2d, bias, *flat)
    return y.reshape(xs[0].shape[0], weight2d.shape[0], *xs[0].shape[2:])


def _compute_dtype(x, *ws):
    if x.is_cuda and torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
    else:
        dt = x.dtype
    return (x.to(dt),) + ws                                 # the weights stay masters: the Functions make their own copies


def linear_cl(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """F.linear with the split-K weight gradient; follows autocast like F.linear."""
    x, weight, bias = _compute_dtype(x, weight, bias)
    return _LinearCL.apply(x, weight, bias)


def pointwise(x: torch.Tensor, weight2d: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """x (B, Cin, *spatial) in any memory order, weight2d (Cout, Cin) -> (B, Cout, *spatial) channel-first."""
    x, weight2d, bias = _compute_dtype(x, weight2d, bias)
    B, C = x.shape[:2]
    xs = x.flatten(2)
    if xs.stride(2) != 1 and xs.stride(1) != 1:
        xs = xs.contiguous()
    y = _Pointwise.apply(xs, weight2d, bias)
    return y.reshape(B, weight2d.shape[0], *x.shape[2:])
