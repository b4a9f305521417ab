"""SegMamba network: Mamba encoder + UNETR-style conv decoder.

Host-side mirror of reference model_segmamba/segmamba.py:
  MambaLayer :49-76, MlpChannel :78-89, GSC :91-132, MambaEncoder :134-193,
  SegMamba :195-343.
Constructor signature, forward semantics and the 291 state_dict keys are kept
so `0_inference.py` / `3_train.py` and reference checkpoints work unchanged.

What differs from the reference is only *how* the tensors move:
  * a MambaLayer hands the tri-directional mixer a (B, L, C) channel-last
    view and gets one back - the layout the MI355X scan kernels are written
    for (one lane per channel), so no transposing copy happens on the way in;
  * InstanceNorm + activation (+ residual) chains go through
    `fused_norm.instance_norm_act`.
"""
from __future__ import annotations

import torch.nn as nn

from . import fused_norm, layout
from .conv3d import conv3d_same
from .mamba_simple import Mamba
from .unet_blocks import UnetOutBlock, UnetrBasicBlock, UnetrUpBlock


class MambaLayer(nn.Module):
    """LayerNorm -> Mamba(v3) over the flattened volume -> + skip.  reference: segmamba.py:49-76."""

    def __init__(self, dim, d_state=16, d_conv=4, expand=2, num_slices=None):
        super().__init__()
        self.dim = dim
        self.norm = nn.LayerNorm(dim)
        self.mamba = Mamba(d_model=dim, d_state=d_state, d_conv=d_conv, expand=expand,
                           bimamba_type="v3", nslices=num_slices)

    def forward(self, x):
        B, C = x.shape[:2]
        assert C == self.dim
        img_dims = x.shape[2:]
        n_tokens = img_dims.numel()
        # (B, L, C) LayerNorm'ed tokens: the transpose and the normalisation are one kernel
        tokens = layout.volume_to_tokens_layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        mixed = self.mamba(tokens)                                   # (B, L, C)
        return layout.tokens_to_volume_add(mixed, x)                 # transpose back fused with the skip


class MlpChannel(nn.Module):
    """1x1x1 conv -> GELU -> 1x1x1 conv.  reference: segmamba.py:78-89."""

    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.fc1 = nn.Conv3d(hidden_size, mlp_dim, 1)
        self.act = nn.GELU()
        self.fc2 = nn.Conv3d(mlp_dim, hidden_size, 1)

    def forward(self, x):
        h = self.act(fused_norm.pointwise_conv3d(x, self.fc1.weight, self.fc1.bias))
        return fused_norm.pointwise_conv3d(h, self.fc2.weight, self.fc2.bias)


class GSC(nn.Module):
    """Gated spatial convolution.  reference: segmamba.py:91-132.

    Two branches (3x3x3 -> 3x3x3, and 1x1x1) are *added* (reference :127; the
    paper figure shows a product), projected by a 1x1x1 conv and added to the input.
    """

    def __init__(self, in_channles) -> None:
        super().__init__()
        c = in_channles
        self.proj = nn.Conv3d(c, c, 3, 1, 1)
        self.norm = nn.InstanceNorm3d(c)
        self.nonliner = nn.ReLU()
        self.proj2 = nn.Conv3d(c, c, 3, 1, 1)
        self.norm2 = nn.InstanceNorm3d(c)
        self.nonliner2 = nn.ReLU()
        self.proj3 = nn.Conv3d(c, c, 1, 1, 0)
        self.norm3 = nn.InstanceNorm3d(c)
        self.nonliner3 = nn.ReLU()
        self.proj4 = nn.Conv3d(c, c, 1, 1, 0)
        self.norm4 = nn.InstanceNorm3d(c)
        self.nonliner4 = nn.ReLU()

    def forward(self, x):
        x1, st = conv3d_same(x, self.proj.weight, self.proj.bias, want_stats=True)
        x1 = fused_norm.instance_norm_act(x1, act="relu", eps=self.norm.eps, stats=st)
        x1, st = conv3d_same(x1, self.proj2.weight, self.proj2.bias, want_stats=True)
        x1 = fused_norm.instance_norm_act(x1, act="relu", eps=self.norm2.eps, stats=st)
        x2 = fused_norm.instance_norm_act(fused_norm.pointwise_conv3d(x, self.proj3.weight, self.proj3.bias),
                                          act="relu", eps=self.norm3.eps)
        y = fused_norm.instance_norm_act(fused_norm.pointwise_conv3d(x1 + x2, self.proj4.weight, self.proj4.bias),
                                         act="relu", eps=self.norm4.eps)
        return y + x


class MambaEncoder(nn.Module):
    """reference: segmamba.py:134-193."""

    NUM_SLICES = (64, 32, 16, 8)        # reference :154

    def __init__(self, in_chans=1, depths=[2, 2, 2, 2], dims=[48, 96, 192, 384],
                 drop_path_rate=0., layer_scale_init_value=1e-6, out_indices=[0, 1, 2, 3]):
        super().__init__()
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            nn.Conv3d(in_chans, dims[0], kernel_size=7, stride=2, padding=3)))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(
                nn.InstanceNorm3d(dims[i]),
                nn.Conv3d(dims[i], dims[i + 1], kernel_size=2, stride=2)))

        self.stages = nn.ModuleList()
        self.gscs = nn.ModuleList()
        for i in range(4):
            self.gscs.append(GSC(dims[i]))
            self.stages.append(nn.Sequential(
                *[MambaLayer(dim=dims[i], num_slices=self.NUM_SLICES[i]) for _ in range(depths[i])]))

        self.out_indices = out_indices
        self.mlps = nn.ModuleList()
        for i in range(4):
            self.add_module(f"norm{i}", nn.InstanceNorm3d(dims[i]))
            self.mlps.append(MlpChannel(dims[i], 2 * dims[i]))

    def forward_features(self, x):
        outs = []
        for i in range(4):
            if i == 0:
                stem = self.downsample_layers[0][0]
                x = fused_norm.stem_conv3d(x, stem.weight, stem.bias)
            else:                                           # InstanceNorm -> conv k2 s2 (a GEMM on 2x2x2 patches)
                norm, conv = self.downsample_layers[i][0], self.downsample_layers[i][1]
                x = fused_norm.instance_norm_act(x, act="none", eps=norm.eps)
                if all(s % 2 == 0 for s in x.shape[2:]):
                    x = fused_norm.patch_conv3d(x, conv.weight, conv.bias, 2)
                else:
                    x = conv(x)
            x = self.gscs[i](x)
            x = self.stages[i](x)
            if i in self.out_indices:
                norm = getattr(self, f"norm{i}")
                outs.append(self.mlps[i](fused_norm.instance_norm_act(x, act="none", eps=norm.eps)))
        return tuple(outs)

    def forward(self, x):
        return self.forward_features(x)


class SegMamba(nn.Module):
    """reference: segmamba.py:195-343."""

    def __init__(self, in_chans=1, out_chans=13, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384],
                 drop_path_rate=0, layer_scale_init_value=1e-6, hidden_size: int = 768,
                 norm_name="instance", conv_block: bool = True, res_block: bool = True,
                 spatial_dims=3) -> None:
        super().__init__()
        self.hidden_size = hidden_size
        self.in_chans = in_chans
        self.out_chans = out_chans
        self.depths = depths
        self.drop_path_rate = drop_path_rate
        self.feat_size = feat_size
        self.layer_scale_init_value = layer_scale_init_value
        self.spatial_dims = spatial_dims

        self.vit = MambaEncoder(in_chans, depths=depths, dims=feat_size, drop_path_rate=drop_path_rate,
                                layer_scale_init_value=layer_scale_init_value)

        def enc(cin, cout):
            return UnetrBasicBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout,
                                   kernel_size=3, stride=1, norm_name=norm_name, res_block=res_block)

        def dec(cin, cout):
            return UnetrUpBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, kernel_size=3,
                                upsample_kernel_size=2, norm_name=norm_name, res_block=res_block)

        f = self.feat_size
        self.encoder1 = enc(self.in_chans, f[0])
        self.encoder2 = enc(f[0], f[1])
        self.encoder3 = enc(f[1], f[2])
        self.encoder4 = enc(f[2], f[3])
        self.encoder5 = enc(f[3], self.hidden_size)
        self.decoder5 = dec(self.hidden_size, f[3])
        self.decoder4 = dec(f[3], f[2])
        self.decoder3 = dec(f[2], f[1])
        self.decoder2 = dec(f[1], f[0])
        self.decoder1 = enc(f[0], f[0])
        self.out = UnetOutBlock(spatial_dims=spatial_dims, in_channels=48, out_channels=self.out_chans)

    def forward(self, x_in):
        outs = self.vit(x_in)
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(outs[0])
        enc3 = self.encoder3(outs[1])
        enc4 = self.encoder4(outs[2])
        enc_hidden = self.encoder5(outs[3])
        dec3 = self.decoder5(enc_hidden, enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        return self.out(self.decoder1(dec0))
