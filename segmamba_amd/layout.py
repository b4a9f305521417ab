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


def transpose_add(x: torch.Tensor, add: torch.Tensor | None = None) -> torch.Tensor:
    """x (B, R, C) -> (B, C, R) contiguous, plus `add` (B, C, R) if given."""
    if not x.is_cuda:
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
