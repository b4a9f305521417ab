"""3D-conv encoder/decoder blocks of the SegMamba stem.

Host-side mirror of the three MONAI blocks the reference model instantiates
(reference: monai/networks/blocks/dynunet_block.py:25-111 `UnetResBlock`,
:247-267 `UnetOutBlock`, :270-312 `get_conv_layer`/`get_padding`;
monai/networks/blocks/unetr_block.py:22-86 `UnetrUpBlock`, :209-260
`UnetrBasicBlock`; monai/networks/blocks/convolutions.py:150-170 `Convolution`).

Only the configuration SegMamba uses is built: 3 spatial dims, instance norm
(no affine, eps 1e-5), LeakyReLU(0.01), residual blocks, no dropout, no conv
bias except in the output head.  Parameter names are kept identical
(`<block>.layer.conv1.conv.weight`, `<block>.transp_conv.conv.weight`,
`out.conv.conv.{weight,bias}` ...) so reference checkpoints load unchanged.

The dense convolutions run on MIOpen through `torch.nn`;
the normalisation + activation (+ residual) epilogues go through
`segmamba_amd.fused_norm` which uses the hand-written HIP kernels when the
tensors are on the GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import fused_norm
from . import conv3d as _conv3d
from .conv3d import conv3d_same, conv3d_same_cat


def _same_padding(kernel_size: int, stride: int) -> int:
    # reference: dynunet_block.py:304-312 -> (k - s + 1) / 2, must be >= 0
    pad2 = kernel_size - stride + 1
    if pad2 < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    return pad2 // 2


def _transposed_output_padding(kernel_size: int, stride: int, padding: int) -> int:
    # reference: dynunet_block.py:315-327
    out_pad = 2 * padding + stride - kernel_size
    if out_pad < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    return out_pad


class ConvOnly(nn.Sequential):
    """A bare convolution registered under the child name ``conv``.

    Mirrors what MONAI's `Convolution` degenerates to when neither activation,
    normalisation nor dropout is requested (convolutions.py:155-160): a
    `Sequential` holding exactly one module called "conv".
    """

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 bias: bool = False, transposed: bool = False):
        super().__init__()
        pad = _same_padding(kernel_size, stride)
        if transposed:
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                      padding=pad,
                                      output_padding=_transposed_output_padding(kernel_size, stride, pad),
                                      bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                             padding=pad, bias=bias)
        self.add_module("conv", conv)
        self._pointwise = (not transposed) and kernel_size == 1 and stride == 1
        self._patch = kernel_size == stride and kernel_size > 1 and pad == 0      # non-overlapping: a GEMM
        self._k = kernel_size
        self._transposed = transposed
        self._same = (not transposed) and stride == 1 and kernel_size == 3

    def forward(self, x, want_stats: bool = False):
        """`x` may be a tuple of tensors standing for their channel concatenation (never materialised).
        want_stats: -> (y, stats) with the InstanceNorm partials the convolution summed in its epilogue, or None (conv3d.py)."""
        if want_stats:
            if self._same and not (not isinstance(x, (tuple, list)) and self.conv.weight.shape[1] <= 4):
                if isinstance(x, (tuple, list)):
                    if self.conv.bias is None:
                        return conv3d_same_cat(tuple(x), self.conv.weight, want_stats=True)
                else:
                    return conv3d_same(x, self.conv.weight, self.conv.bias, want_stats=True)
            return self.forward(x), None
        if isinstance(x, (tuple, list)):
            weight = self.conv.weight                    # sliced per part: the Functions copy the slices they are given
            if self._same and self.conv.bias is None:
                return conv3d_same_cat(tuple(x), weight)
            if self._pointwise:
                from . import lib as L, linear
                if _conv3d._CAT_FUSED and all(L.on_device(part) for part in x):      # one node, the parts added in place
                    out = linear.pointwise_cat(tuple(x), weight.reshape(weight.shape[0], weight.shape[1]), self.conv.bias)
                    if out is not None:
                        return out
                out, c0 = None, 0
                for part in x:
                    c = part.shape[1]
                    y = fused_norm.pointwise_conv3d(part, weight[:, c0:c0 + c],
                                                    self.conv.bias if c0 == 0 else None)
                    out = y if out is None else out + y
                    c0 += c
                return out
            x = torch.cat(tuple(x), dim=1)
        if self._pointwise:
            return fused_norm.pointwise_conv3d(x, self.conv.weight, self.conv.bias)
        if self._same:
            if self.conv.weight.shape[1] <= 4:           # the network's first layer: 4 input channels, not 48 with 44 zero ones
                y = fused_norm.thin_conv3d_same(x, self.conv.weight, self.conv.bias)
                if y is not None:
                    return y
            return conv3d_same(x, self.conv.weight, self.conv.bias)
        if self._patch and all(s % self._k == 0 for s in x.shape[2:]):
            if self._transposed:
                return fused_norm.patch_conv_transpose3d(x, self.conv.weight, self.conv.bias, self._k)
            return fused_norm.patch_conv3d(x, self.conv.weight, self.conv.bias, self._k)
        return self.conv(x)


class UnetResBlock(nn.Module):
    """conv3 -> IN -> LeakyReLU -> conv3 -> IN  (+ [1x1 conv -> IN] skip) -> add -> LeakyReLU.

    reference: dynunet_block.py:44-111.  `norm{1,2,3}` / `lrelu` attributes are
    kept (parameter-free) so `named_modules()` lines up with the reference.
    """

    NEG_SLOPE = 0.01

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1):
        super().__init__()
        self.conv1 = ConvOnly(in_channels, out_channels, kernel_size, stride)
        self.conv2 = ConvOnly(out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=self.NEG_SLOPE, inplace=True)
        self.norm1 = nn.InstanceNorm3d(out_channels)
        self.norm2 = nn.InstanceNorm3d(out_channels)
        self.downsample = (in_channels != out_channels) or stride != 1
        if self.downsample:
            self.conv3 = ConvOnly(in_channels, out_channels, 1, stride)
            self.norm3 = nn.InstanceNorm3d(out_channels)

    def forward(self, inp) -> torch.Tensor:
        """`inp`: a tensor, or a tuple of tensors meaning their channel concatenation (decoder: (upsampled, skip))."""
        skip = None
        if self.downsample and self.conv1._same and self.conv3._pointwise and self.conv1.conv.bias is None and self.conv3.conv.bias is None:
            front = _conv3d.res_front(tuple(inp) if isinstance(inp, (tuple, list)) else (inp,), self.conv1.conv.weight,
                                      self.conv3.conv.weight, want_stats=True)
            if front is not None:
                out, st, skip = front
        if skip is None:
            out, st = self.conv1(inp, want_stats=True)
        out = fused_norm.instance_norm_act(out, act="leaky_relu", slope=self.NEG_SLOPE, eps=self.norm1.eps, stats=st)
        out, st = self.conv2(out, want_stats=True)
        if self.downsample:
            residual = fused_norm.instance_norm_act(self.conv3(inp) if skip is None else skip, act="none", eps=self.norm3.eps)
        else:
            residual = torch.cat(tuple(inp), dim=1) if isinstance(inp, (tuple, list)) else inp
        # IN(out) + residual -> LeakyReLU, one pass
        return fused_norm.instance_norm_act(out, act="leaky_relu", slope=self.NEG_SLOPE, eps=self.norm2.eps,
                                            residual=residual, stats=st)


class UnetrBasicBlock(nn.Module):
    """reference: unetr_block.py:209-260 (res_block=True branch only)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size: int, stride: int,
                 norm_name="instance", res_block: bool = True):
        super().__init__()
        _check_config(spatial_dims, norm_name, res_block)
        self.layer = UnetResBlock(in_channels, out_channels, kernel_size, stride)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        return self.layer(inp)


class UnetrUpBlock(nn.Module):
    """ConvTranspose3d(k=s=upsample) -> cat(skip) -> UnetResBlock.  reference: unetr_block.py:22-86."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size: int,
                 upsample_kernel_size: int, norm_name="instance", res_block: bool = True):
        super().__init__()
        _check_config(spatial_dims, norm_name, res_block)
        self.transp_conv = ConvOnly(in_channels, out_channels, upsample_kernel_size, upsample_kernel_size,
                                    transposed=True)
        self.conv_block = UnetResBlock(2 * out_channels, out_channels, kernel_size, 1)

    def forward(self, inp: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
        out = self.transp_conv(inp)
        return self.conv_block((out, skip))               # cat(out, skip) is never materialised


class UnetOutBlock(nn.Module):
    """1x1x1 conv with bias.  reference: dynunet_block.py:247-267."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int):
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("SegMamba hot path is 3-D only")
        self.conv = ConvOnly(in_channels, out_channels, 1, 1, bias=True)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        return self.conv(inp)


def _check_config(spatial_dims, norm_name, res_block):
    if spatial_dims != 3:
        raise NotImplementedError("SegMamba hot path is 3-D only")
    name = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
    if str(name).lower() != "instance":
        raise NotImplementedError(f"norm {norm_name!r}: only instance norm is on the SegMamba path")
    if not res_block:
        raise NotImplementedError("res_block=False is not on the SegMamba path")
