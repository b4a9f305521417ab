"""Channel-first volume <-> channel-last token moves around a Mamba layer (reference segmamba.py:60-75).

On the GPU both directions are the library's LDS-tiled transpose (csrc/layout.hip, segm_transpose_add), the way back
fused with the skip connection; each is the other's backward.  CPU tensors take the ATen expression.
"""
from __future__ import annotations

import torch


class _TransposeAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, add):
        from . import lib as L, ops_raw
        ctx.has_add = add is not None
        return ops_raw.transpose_add(L.get_lib(), x.contiguous(), add.contiguous() if add is not None else None)

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L, ops_raw
        dout = dout.contiguous()
        dx = ops_raw.transpose_add(L.get_lib(), dout) if ctx.needs_input_grad[0] else None
        return dx, (dout if ctx.has_add and ctx.needs_input_grad[1] else None)


class _TokensLayerNorm(torch.autograd.Function):
    """(B, C, S) volume -> LayerNorm'ed (B, S, C) tokens through segm_layernorm_tokens_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        from . import lib as L, ops_raw
        y, mean, rstd = ops_raw.layernorm_tokens_fwd(L.get_lib(), x, weight, bias, eps)
        ctx.save_for_backward(x, mean, rstd, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L, ops_raw
        x, mean, rstd, weight = ctx.saved_tensors
        dx, dg, db = ops_raw.layernorm_tokens_bwd(L.get_lib(), x, dy.to(x.dtype), mean, rstd, weight)
        return dx, dg.to(weight.dtype), db.to(weight.dtype), None


def volume_to_tokens_layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm(C)(x.reshape(B, C, n).transpose(-1, -2)) as one kernel per direction (reference segmamba.py:60-66).
    Like autocast's LayerNorm the statistics are fp32; the result is returned in x's dtype."""
    B, C = x.shape[:2]
    x3 = x.reshape(B, C, -1)
    from . import lib as L
    if L.on_device(x):
        from . import ops_raw
        x3 = x3.contiguous()
        if ops_raw.layernorm_tokens_supported(x3):
            return _TokensLayerNorm.apply(x3, weight, bias, eps)
    return torch.nn.functional.layer_norm(transpose_add(x3), (C,), weight, bias, eps)


def transpose_add(x: torch.Tensor, add: torch.Tensor | None = None) -> torch.Tensor:
    """x (B, R, C) -> (B, C, R) contiguous, plus `add` (B, C, R) if given."""
    from . import lib as L
    if not L.on_device(x):
        y = x.transpose(1, 2)
        return (y + add) if add is not None else y.contiguous()
    if add is not None and add.dtype != x.dtype:
        dt = torch.promote_types(x.dtype, add.dtype)
        x, add = x.to(dt), add.to(dt)
    return _TransposeAdd.apply(x, add)


def volume_to_tokens(x: torch.Tensor) -> torch.Tensor:
    """(B, C, *spatial) -> (B, L, C) contiguous tokens."""
    B, C = x.shape[:2]
    return transpose_add(x.reshape(B, C, -1))


def tokens_to_volume_add(tokens: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """(B, L, C) tokens -> skip's (B, C, *spatial) shape, + skip."""
    B, C = skip.shape[:2]
    return transpose_add(tokens, skip.reshape(B, C, -1)).reshape(skip.shape)
