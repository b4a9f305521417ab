"""`mamba_ssm.Mamba` with `bimamba_type="v3"` on the MI355X kernels.

Host-side mirror of reference mamba/mamba_ssm/modules/mamba_simple.py:35-264: same constructor signature,
parameter names / shapes / initialisation (so state_dicts are interchangeable) and the same forward
semantics for the tri-directional mixer SegMamba uses.

Data movement differs from the reference on purpose (MI355X-first):
  * `xz` stays in the (B, L, 2*d_inner) layout the in-projection GEMM produces - the scan / conv kernels
    own one channel per lane, so channel-last rows are their coalesced layout (the reference transposes to
    (B, 2*d_inner, L), :204-208);
  * the reversed and slice-interleaved directions (:231, :245-247, :261, :264) are `time_order` arguments of
    the kernels; nothing is flipped, stacked or permuted in memory;
  * the direction outputs are already (B, L, d_inner), which is what `out_proj` consumes (:264 rearranges).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import lib as L
from .linear import linear_cl
from .selective_scan_interface import _inner


class Mamba(nn.Module):
    def __init__(
        self,
        d_model,
        d_state=16,
        d_conv=4,
        expand=2,
        dt_rank="auto",
        dt_min=0.001,
        dt_max=0.1,
        dt_init="random",
        dt_scale=1.0,
        dt_init_floor=1e-4,
        conv_bias=True,
        bias=False,
        use_fast_path=True,
        layer_idx=None,
        device=None,
        dtype=None,
        bimamba_type="none",
        nslices=5,
    ):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        self.bimamba_type = bimamba_type
        self.nslices = nslices
        assert bimamba_type == "v3"                     # reference :125 - the only constructible variant

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.activation = "silu"
        self.act = nn.SiLU()

        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init not in ("constant", "random"):
            raise NotImplementedError

        # three parameter sets: "" (as stored), "_b" (reversed), "_s" (slice-interleaved); reference :69-186.
        # Creation order follows the reference so that a seeded construction draws the same random numbers.
        def conv():
            return nn.Conv1d(in_channels=self.d_inner, out_channels=self.d_inner, bias=conv_bias, kernel_size=d_conv,
                             groups=self.d_inner, padding=d_conv - 1, **factory_kwargs)

        def x_proj():
            return nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)

        def dt_proj():
            return nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)

        def a_log():
            A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1)
            p = nn.Parameter(torch.log(A).contiguous())          # S4D-real init, kept in fp32
            p._no_weight_decay = True
            return p

        def d_skip():
            p = nn.Parameter(torch.ones(self.d_inner, device=device))   # kept in fp32
            p._no_weight_decay = True
            return p

        self.conv1d = conv()
        self.x_proj = x_proj()
        self.dt_proj = dt_proj()
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        else:
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        # dt bias such that softplus(bias) is log-uniform in [dt_min, dt_max]  (reference :98-108)
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True
        self.A_log = a_log()
        self.D = d_skip()

        self.A_b_log = a_log()
        self.conv1d_b = conv()
        self.x_proj_b = x_proj()
        self.dt_proj_b = dt_proj()
        self.D_b = d_skip()

        self.A_s_log = a_log()
        self.conv1d_s = conv()
        self.x_proj_s = x_proj()
        self.dt_proj_s = dt_proj()
        self.D_s = d_skip()

        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)

    def _direction(self, xz, suffix, time_order, nslices=1):
        conv = getattr(self, "conv1d" + suffix)
        x_proj = getattr(self, "x_proj" + suffix)
        dt_proj = getattr(self, "dt_proj" + suffix)
        A = -torch.exp(getattr(self, "A" + suffix + "_log").float())
        return _inner(xz, conv.weight, conv.bias, x_proj.weight, dt_proj.weight, A, None, None,
                      getattr(self, "D" + suffix).float(), dt_proj.bias.float(), None, None, True,
                      channel_last=True, time_order=time_order, nslices=nslices)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> (B, L, D)"""
        if inference_params is not None:
            raise NotImplementedError("autoregressive decoding is outside the SegMamba hot path (SURVEY.md §2.1)")
        batch, seqlen, _ = hidden_states.shape
        if seqlen % self.nslices != 0:
            raise RuntimeError(f"sequence length {seqlen} must be divisible by nslices {self.nslices}")
        xz = linear_cl(hidden_states, self.in_proj.weight, self.in_proj.bias)        # (B, L, 2*d_inner)
        out = self._direction(xz, "", L.TIME_FORWARD)
        out_b = self._direction(xz, "_b", L.TIME_REVERSED)
        out_s = self._direction(xz, "_s", L.TIME_INTERLEAVED, self.nslices)
        return linear_cl(out + out_b + out_s, self.out_proj.weight, self.out_proj.bias)

    def step(self, hidden_states, conv_state, ssm_state):
        raise NotImplementedError("autoregressive decoding is outside the SegMamba hot path (SURVEY.md §2.1)")

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("autoregressive decoding is outside the SegMamba hot path (SURVEY.md §2.1)")
