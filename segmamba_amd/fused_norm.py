"""InstanceNorm3d (+ residual) (+ activation) epilogues of the conv stem.

reference call sites: model_segmamba/segmamba.py:96-130 (`GSC`: IN -> ReLU),
:147,169-187 (bare IN before the down-sampling conv / the per-stage MLP),
monai/networks/blocks/dynunet_block.py:98-111 (IN -> LeakyReLU, and
IN -> +residual -> LeakyReLU).

`torch.nn.InstanceNorm3d` defaults apply everywhere on the path: no affine
parameters, no running statistics, eps = 1e-5, biased variance.

On the GPU every call is two launches of the library's own kernels per direction (csrc/instnorm.hip:
statistics, then normalise + residual + activation in one pass; the backward likewise) instead of the three to four
ATen kernels per direction the reference runs (batch_norm statistics / transform, add, leaky_relu and their
backward counterparts: 41 ms of a 283 ms training step, profiles/r01_bench_step_kernels_v5.txt).  CPU tensors
(unit tests of the host logic) take the ATen expression of the same math.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import linear


class _InstNormAct(torch.autograd.Function):
    """y = act(IN(x) + residual) through segm_instnorm_fwd / segm_instnorm_bwd."""

    @staticmethod
    def forward(ctx, x, residual, act, slope, eps, stats=None):
        from . import lib as L, ops_raw
        if not ops_raw.channel_dense(x):                 # a convolution output with a padded channel stride is taken as it is
            x = x.contiguous()
        if residual is not None:
            residual = residual.to(x.dtype)
            if not ops_raw.channel_dense(residual):
                residual = residual.contiguous()
        y, mean, rstd = ops_raw.instnorm_fwd(L.get_lib(), x, residual, act, slope, eps, stats=stats)
        need_y = residual is not None and act != "none"
        ctx.save_for_backward(x, mean, rstd, y if need_y else None)
        ctx.cfg = (act, slope, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L, ops_raw
        x, mean, rstd, y = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        dx, dres = ops_raw.instnorm_bwd(L.get_lib(), x, dy, mean, rstd, y, act, slope,
                                        want_dresidual=has_res and ctx.needs_input_grad[1])
        return dx, dres, None, None, None, None


def instance_norm_act(x: torch.Tensor, act: str = "none", slope: float = 0.01, eps: float = 1e-5,
                      residual: torch.Tensor | None = None, stats: torch.Tensor | None = None) -> torch.Tensor:
    """y = act(IN(x) [+ residual]) for x of shape (B, C, D, H, W).

    act in {"none", "relu", "leaky_relu"}.  stats: the partial {count, sum, sum of squares} the convolution that produced x summed
    in its epilogue (conv3d.conv3d_same(..., want_stats=True)): the statistics pass over x is skipped.
    """
    if act not in ("none", "relu", "leaky_relu"):
        raise ValueError(f"unknown activation {act!r}")
    from . import lib as L
    if L.on_device(x):
        return _InstNormAct.apply(x, residual, act, slope, eps, stats)
    y = F.instance_norm(x, eps=eps)
    if residual is not None:
        y = y + residual
    if act == "none":
        return y
    if act == "relu":
        return F.relu(y)
    if act == "leaky_relu":
        return F.leaky_relu(y, negative_slope=slope)
    raise ValueError(f"unknown activation {act!r}")


def pointwise_conv3d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """1x1x1 convolution as the GEMM it is (north star: "MFMA only for the 1x1x1 projection GEMMs").

    x (B, C, D, H, W), weight (Cout, C, 1, 1, 1).  MIOpen's conv path is badly tuned for these shapes in bf16 (the
    48 -> 4 output head's weight-gradient kernel alone took 533 ms of a 983 ms training step, profiles/r01_*):
    W (Cout, C) x (C, D*H*W) per volume through the BLAS path takes well under a millisecond.  The GEMM reads x in
    whichever memory order it has (channel-first or channel-last) and writes channel-first; the weight gradient is a
    split-K product (linear.py).
    """
    cout, cin = weight.shape[0], weight.shape[1]
    return linear.pointwise(x, weight.reshape(cout, cin), bias)


class _StemConv(torch.autograd.Function):
    """A convolution on at most 4 input channels through the thin-input kernels (csrc/stem.hip): Conv3d(kernel 7, stride 2,
    padding 3) of the encoder's stem (model_segmamba/segmamba.py:141) and Conv3d(kernel 3, stride 1, padding 1) of `encoder1`'s
    first layer (:236-244, monai dynunet_block.py:72-80) - the geometry follows the weight's shape.  Forward: segm_stem_conv_fwd
    (an implicit GEMM; MIOpen's im2col route took 1.37 ms for the stem on the 2 x 4 x 128^3 input, the 48-channel 3x3x3 kernel
    0.6 ms for the 3^3 layer with 44 of its 48 input channels zero); weight gradient: segm_stem_conv_wgrad on the channel-last-4
    copy of the input the forward made (MIOpen: 3.0 ms for the stem); bias gradient: a sum.  The network input needs no gradient;
    when it does ask for one (or the width is not one the kernel takes) ATen's convolution_backward runs on the same operands."""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import lib as L, ops_raw
        from .linear import _masters
        w, b = _masters(ctx, x, w, b)                    # fp32 masters -> the step's 16-bit copies; gradients go back in fp32
        ctx.k, ctx.s = ops_raw._stem_geometry(w)
        x4 = ops_raw.stem_channel_last4(x)
        ctx.use_hip = (_STEM_WGRAD_HIP and not ctx.needs_input_grad[0] and ops_raw.stem_wgrad_supported(x4, w.shape[0], ctx.k))
        ctx.save_for_backward(x4 if ctx.use_hip else x, w)
        return ops_raw.stem_conv_fwd(L.get_lib(), x, w, b, x4=x4)

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L, ops_raw
        x, w = ctx.saved_tensors
        if ctx.use_hip:
            dw = ops_raw.stem_conv_wgrad(L.get_lib(), x, dy, w.shape[1], ctx.k).to(ctx.w_dtype) if ctx.needs_input_grad[1] else None
            db = linear.bias_grad(dy).to(ctx.b_dtype) if ctx.has_bias and ctx.needs_input_grad[2] else None
            return None, dw, db
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]]
        dx, dw, db = torch.ops.aten.convolution_backward(dy.contiguous(), x, w, [w.shape[0]] if ctx.has_bias else None, [ctx.s] * 3,
                                                         [ctx.k // 2] * 3, [1, 1, 1], False, [0, 0, 0], 1, mask)
        return dx, (dw.to(ctx.w_dtype) if dw is not None else None), (db.to(ctx.b_dtype) if db is not None else None)


_STEM_WGRAD_HIP = os.environ.get("SEGM_STEM_WGRAD_HIP", "1") == "1"
_STEM_HIP = os.environ.get("SEGM_STEM_HIP", "1") == "1"


def thin_conv3d_same(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor | None:
    """Conv3d(kernel 3, stride 1, padding 1) on at most 4 input channels through the thin-input kernels, or None when the call is
    not theirs (SEGM_THIN_CONV_HIP=0, CPU tensors, a 32-bit compute dtype, an input that needs its gradient, other shapes)"""
    from . import lib as L, ops_raw
    if tuple(weight.shape[2:]) != (3, 3, 3):             # the thin kernels infer (kernel, stride) from the weight: 7^3 would run stride 2
        return None
    if not (_THIN_HIP and L.on_device(x) and weight.shape[1] <= 4 and not x.requires_grad and ops_raw.stem_conv_supported(x, weight)):
        return None
    dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
    if dt not in (torch.bfloat16, torch.float16):
        return None
    return _StemConv.apply(x.to(dt), weight, bias)


_THIN_HIP = os.environ.get("SEGM_THIN_CONV_HIP", "1") == "1"


def stem_conv3d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """The 7^3 stride-2 stem convolution; follows autocast like F.conv3d.  Library kernel for 16-bit activations on the device
    (SEGM_STEM_HIP=0: the MIOpen call), F.conv3d otherwise."""
    from . import lib as L, ops_raw
    if tuple(weight.shape[2:]) != (7, 7, 7):
        raise RuntimeError(f"stem_conv3d: a 7x7x7 stride-2 convolution is expected, got a kernel of {tuple(weight.shape[2:])}")
    if _STEM_HIP and L.on_device(x) and ops_raw.stem_conv_supported(x, weight):
        dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
        if dt in (torch.bfloat16, torch.float16):
            return _StemConv.apply(x.to(dt), weight, bias)             # the weights stay masters
    return F.conv3d(x, weight, bias, stride=2, padding=3)


def patch_conv3d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, k: int) -> torch.Tensor:
    """Conv3d with kernel_size == stride == k and no padding (SegMamba's down-sampling convs, segmamba.py:148): the
    patches do not overlap, so it is one GEMM on a space-to-depth view.  x (B, C, D, H, W), weight (Cout, C, k, k, k)."""
    B, C, D, H, W = x.shape
    cout = weight.shape[0]
    from . import lib as L
    if _D2S_HIP and k == 2 and L.on_device(x) and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 5 and W % 16 == 0 \
            and D % 2 == 0 and H % 2 == 0 and x.stride(4) == 1 and all(s % 8 == 0 for s in x.stride()[:4]) and x.data_ptr() % 16 == 0:
        blk = _SpaceToDepth2.apply(x)                                                          # (B, C*8, d, h, w) channel-first, one kernel
        return linear.pointwise(blk, weight.reshape(cout, C * 8), bias)
    xp = x.reshape(B, C, D // k, k, H // k, k, W // k, k).permute(0, 2, 4, 6, 1, 3, 5, 7)     # (B, d, h, w, C, k, k, k)
    xp = xp.reshape(B, (D // k) * (H // k) * (W // k), C * k ** 3)                             # the one gather copy
    y = linear.pointwise(xp.transpose(1, 2), weight.reshape(cout, C * k ** 3), bias)          # (B, Cout, d*h*w)
    return y.reshape(B, cout, D // k, H // k, W // k)


class _DepthToSpace2(torch.autograd.Function):
    """(B, C * 8, D, H, W) -> (B, C, 2D, 2H, 2W), vol[b, c, 2z+i, 2y+j, 2x+k] = blk[b, c, i, j, k, z, y, x] through
    segm_depth_to_space2 (16-byte accesses on both sides; ATen's strided copy moved these at ~2 TB/s); the backward is the inverse
    gather through the same kernel."""

    @staticmethod
    def forward(ctx, blk):
        from . import lib as L, ops_raw
        return ops_raw.depth_to_space2(L.get_lib(), blk)

    @staticmethod
    def backward(ctx, dvol):
        from . import lib as L, ops_raw
        if dvol.stride(4) != 1 or any(s % 8 for s in dvol.stride()[:4]) or dvol.data_ptr() % 16:
            dvol = dvol.contiguous()
        return ops_raw.space_to_depth2(L.get_lib(), dvol)


class _SpaceToDepth2(torch.autograd.Function):
    """the inverse: (B, C, 2D, 2H, 2W) -> (B, C * 8, D, H, W) for the kernel-2 stride-2 down-sampling convolutions (reference
    model_segmamba/segmamba.py:145-150); backward = depth-to-space of the gradient"""

    @staticmethod
    def forward(ctx, vol):
        from . import lib as L, ops_raw
        return ops_raw.space_to_depth2(L.get_lib(), vol)

    @staticmethod
    def backward(ctx, dblk):
        from . import lib as L, ops_raw
        return ops_raw.depth_to_space2(L.get_lib(), dblk.contiguous())


_D2S_HIP = os.environ.get("SEGM_D2S_HIP", "1") == "1"


def patch_conv_transpose3d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, k: int) -> torch.Tensor:
    """ConvTranspose3d with kernel_size == stride == k, no padding (the UNETR up-sampling, unetr_block.py:52-60):
    every input voxel writes its own k^3 output block, so it is one GEMM followed by a depth-to-space permute.
    x (B, C, D, H, W), weight (C, Cout, k, k, k).  (MIOpen's kernels for these shapes took 0.6 s per training step,
    profiles/r01_bench_step_kernels_v2.txt.)"""
    B, C, D, H, W = x.shape
    cout = weight.shape[1]
    y = linear.pointwise(x, weight.reshape(C, cout * k ** 3).t())                              # (B, Cout*k^3, D, H, W)
    from . import ops_raw
    if _D2S_HIP and k == 2 and ops_raw.depth_to_space2_supported(y):
        y = _DepthToSpace2.apply(y)                                                            # the permute as one library kernel
    else:
        y = y.reshape(B, cout, k, k, k, D, H, W).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, cout, D * k, H * k, W * k)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1, 1)
    return y
