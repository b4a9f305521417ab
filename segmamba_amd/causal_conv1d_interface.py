"""`causal_conv1d` Python API on the MI355X kernels.

Host-side mirror of reference causal-conv1d/causal_conv1d/causal_conv1d_interface.py:10-104:
same names, argument meaning, dtype rules and error behaviour; the native calls
`causal_conv1d_cuda.causal_conv1d_fwd/bwd` are replaced by `segm_causal_conv1d_fwd/bwd`
from libsegmamba_hip.so.  Tensors keep the reference's (batch, dim, seqlen) layout here; the kernels
take strides, so channel-last views ((B, L, D) transposed, stride(1) == 1 - the layout the
reference calls "channel_last", causal_conv1d.cpp:152) run on the fast coalesced path without a copy.
"""
from __future__ import annotations

import torch

from . import lib as L
from . import ops_raw


class CausalConv1dFn(torch.autograd.Function):
    """reference: causal_conv1d_interface.py:10-34"""

    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in [None, "silu", "swish"]:
            raise NotImplementedError("activation must be None, silu, or swish")
        if x.stride(2) != 1 and x.stride(1) != 1:
            x = x.contiguous()
        bias = bias.contiguous() if bias is not None else None
        ctx.save_for_backward(x, weight, bias)
        ctx.activation = activation in ["silu", "swish"]
        w32 = weight.float().contiguous()
        b32 = bias.float() if bias is not None else None
        return ops_raw.conv1d_fwd(L.get_lib(), x, w32, b32, ctx.activation, channel_last=False)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(2) != 1 and dout.stride(1) != 1:
            dout = dout.contiguous()
        w32 = weight.float().contiguous()
        b32 = bias.float() if bias is not None else None
        dx, dweight, dbias = ops_raw.conv1d_bwd(L.get_lib(), x, w32, b32, dout, ctx.activation, channel_last=False)
        dweight = dweight.to(weight.dtype)
        dbias = dbias.to(bias.dtype) if bias is not None else None
        return dx, dweight, dbias, None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """
    x: (batch, dim, seqlen)      weight: (dim, width)      bias: (dim,)
    activation: either None or "silu" or "swish"
    out: (batch, dim, seqlen)
    """
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """Single-token decode step (reference causal_conv1d_interface.py:68-82; native op causal_conv1d.cpp:270-330).

    x: (batch, dim)    conv_state: (batch, dim, width), updated in place    weight: (dim, width)    bias: (dim,)
    out: (batch, dim)
    """
    if activation not in [None, "silu", "swish"]:
        raise NotImplementedError("activation must be None, silu, or swish")
    w32 = weight.float().contiguous()
    b32 = bias.float().contiguous() if bias is not None else None
    return ops_raw.conv1d_update(L.get_lib(), x, conv_state, w32, b32, activation in ["silu", "swish"])
