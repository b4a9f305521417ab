"""`causal_conv1d_cuda` - stands in for the pybind module of causal-conv1d/csrc/causal_conv1d.cpp:329-333.

    causal_conv1d_fwd(x, weight, bias, silu) -> out                                  causal_conv1d.cpp:130-189
    causal_conv1d_bwd(x, weight, bias, dout, dx, silu) -> [dx, dweight, dbias]       causal_conv1d.cpp:191-268 (dx may be pre-allocated)
    causal_conv1d_update(x, conv_state, weight, bias, silu) -> out                   causal_conv1d.cpp:270-327 (state updated in place)

x (B, D, L) with stride(-1) == 1 (or channel-last, stride(1) == 1); weight (D, W), W in [2, 4]; weight / bias in any of
fp32 / fp16 / bf16 (causal_conv1d.cpp:136-137) - the kernels take them as fp32.
"""
import torch

from .. import lib as L
from .. import ops_raw


def _w(t):
    return None if t is None else t.float().contiguous()


def causal_conv1d_fwd(x, weight, bias, silu_activation):
    channel_last = x.stride(1) == 1 and x.stride(2) != 1
    if channel_last:
        return ops_raw.conv1d_fwd(L.get_lib(), x.transpose(1, 2), _w(weight), _w(bias), bool(silu_activation),
                                  channel_last=True).transpose(1, 2)
    return ops_raw.conv1d_fwd(L.get_lib(), x, _w(weight), _w(bias), bool(silu_activation))


def causal_conv1d_bwd(x, weight, bias, dout, dx, silu_activation):
    channel_last = x.stride(1) == 1 and x.stride(2) != 1
    if channel_last:
        dx_, dw, db = ops_raw.conv1d_bwd(L.get_lib(), x.transpose(1, 2), _w(weight), _w(bias), dout.transpose(1, 2),
                                         bool(silu_activation), channel_last=True,
                                         dx=None if dx is None else dx.transpose(1, 2))
        dx_ = dx_.transpose(1, 2)
    else:
        dx_, dw, db = ops_raw.conv1d_bwd(L.get_lib(), x, _w(weight), _w(bias), dout, bool(silu_activation), dx=dx)
    return [dx_, dw.to(weight.dtype), db.to(bias.dtype) if bias is not None else torch.zeros_like(dw[:, 0])]


def causal_conv1d_update(x, conv_state, weight, bias, silu_activation):
    return ops_raw.conv1d_update(L.get_lib(), x, conv_state, _w(weight), _w(bias), bool(silu_activation))
