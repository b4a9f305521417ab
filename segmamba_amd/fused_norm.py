"""InstanceNorm3d (+ residual) (+ activation) epilogues of the conv stem.

reference call sites: model_segmamba/segmamba.py:96-130 (`GSC`: IN -> ReLU),
:147,169-187 (bare IN before the down-sampling conv / the per-stage MLP),
monai/networks/blocks/dynunet_block.py:98-111 (IN -> LeakyReLU, and
IN -> +residual -> LeakyReLU).

`torch.nn.InstanceNorm3d` defaults apply everywhere on the path: no affine
parameters, no running statistics, eps = 1e-5, biased variance.

Round 1: the math is expressed with ATen ops (SURVEY.md §8f rank 1 lists the
hand-written fused kernel as the first "next" item once rows a-e are done).
Keeping every call site behind this one function means the HIP kernel drops
in here without touching the model code.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def instance_norm_act(x: torch.Tensor, act: str = "none", slope: float = 0.01, eps: float = 1e-5,
                      residual: torch.Tensor | None = None) -> torch.Tensor:
    """y = act(IN(x) [+ residual]) for x of shape (B, C, D, H, W).

    act in {"none", "relu", "leaky_relu"}.
    """
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
    a (B*D*H*W, C) x (C, Cout) matrix product through the BLAS path takes well under a millisecond.  With
    channels_last_3d activations the permutes below are views, otherwise they cost one transposing copy each way.
    """
    cout, cin = weight.shape[0], weight.shape[1]
    y = F.linear(x.permute(0, 2, 3, 4, 1), weight.reshape(cout, cin), bias)       # (B, D, H, W, Cout)
    return y.permute(0, 4, 1, 2, 3)
