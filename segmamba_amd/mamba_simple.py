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
import os

import torch
import torch.nn as nn

from . import lib as L
from .linear import linear_cl
from .selective_scan_interface import MambaInnerCore3, _inner


_FUSED3 = os.environ.get("SEGM_MAMBA_FUSED3", "1") == "1"     # 0: one autograd node per direction (round 2)


class _Sum3(torch.autograd.Function):
    """out + out_b + out_s (reference mamba_simple.py:160 / :264) as one pass over the three tensors (segm_add3: fp32 sum, one
    rounding) instead of two binary adds; every operand receives the incoming gradient unchanged."""

    @staticmethod
    def forward(ctx, a, b, c):
        from . import ops_raw
        return ops_raw.add3(L.get_lib(), a, b, c)

    @staticmethod
    def backward(ctx, g):
        return g, g, g


class _NegExp3(torch.autograd.Function):
    """A = -exp(A_log) for the three directions of a layer (reference mamba_simple.py:209 / :232 / :248) as two multi-tensor
    launches, and its backward dA_log = dA * A as one - ATen's per-tensor exp / neg and their autograd nodes were twelve launches
    per layer and step (96 of the step's ~190 parameter-sized element-wise launches).  Same values: (-g) * exp(A_log) == g * A."""

    @staticmethod
    def forward(ctx, a_log, b_log, s_log):
        outs = torch._foreach_neg(torch._foreach_exp([a_log.float(), b_log.float(), s_log.float()]))
        ctx.save_for_backward(*outs)
        ctx.dtypes = (a_log.dtype, b_log.dtype, s_log.dtype)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        outs = ctx.saved_tensors
        if any(g is None for g in gs):
            return tuple(None if g is None else (g * o).to(dt) for g, o, dt in zip(gs, outs, ctx.dtypes))
        return tuple(r.to(dt) for r, dt in zip(torch._foreach_mul(list(gs), list(outs)), ctx.dtypes))


def _sum3(a, b, c):
    from . import ops_raw
    from .selective_scan_interface import _ADD3
    if _ADD3 and L.on_device(a) and ops_raw.add3_supported(a, b, c):
        return _Sum3.apply(a, b, c)
    return a + b + c


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
            return self._forward_with_cache(hidden_states, inference_params)
        batch, seqlen, _ = hidden_states.shape
        if seqlen % self.nslices != 0:
            raise RuntimeError(f"sequence length {seqlen} must be divisible by nslices {self.nslices}")
        xz = linear_cl(hidden_states, self.in_proj.weight, self.in_proj.bias)        # (B, L, 2*d_inner)
        if _FUSED3 and L.on_device(xz):
            # the three directions as one autograd node: their scans share one grid (selective_scan_interface.MambaInnerCore3)
            params = []
            As = _NegExp3.apply(self.A_log, self.A_b_log, self.A_s_log)
            for sfx, A in zip(("", "_b", "_s"), As):
                conv, dt_proj = getattr(self, "conv1d" + sfx), getattr(self, "dt_proj" + sfx)
                params += [conv.weight, conv.bias, getattr(self, "x_proj" + sfx).weight, dt_proj.weight,
                           A, getattr(self, "D" + sfx).float(), dt_proj.bias.float()]
            train = torch.is_grad_enabled() and (xz.requires_grad or any(p.requires_grad for p in params))
            out, out_b, out_s = MambaInnerCore3.apply(xz, self.nslices, train, *params)
        else:
            out = self._direction(xz, "", L.TIME_FORWARD)
            out_b = self._direction(xz, "_b", L.TIME_REVERSED)
            out_s = self._direction(xz, "_s", L.TIME_INTERLEAVED, self.nslices)
        return linear_cl(_sum3(out, out_b, out_s), self.out_proj.weight, self.out_proj.bias)

    # ---- autoregressive decoding (reference :196-201, :265-310, :356-436).  SegMamba never takes this path; with
    # `inference_params` the reference runs its uni-directional branch on the forward-direction parameters only. -------------
    def _forward_with_cache(self, hidden_states, inference_params):
        from .causal_conv1d_interface import causal_conv1d_fn
        from .selective_scan_interface import selective_scan_fn
        batch, seqlen, _ = hidden_states.shape
        conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
        if inference_params.seqlen_offset > 0:
            out, _, _ = self.step(hidden_states, conv_state, ssm_state)              # the states are updated in place
            return out
        xz = linear_cl(hidden_states, self.in_proj.weight, self.in_proj.bias).transpose(1, 2)     # (B, 2*d_inner, L)
        A = -torch.exp(self.A_log.float())
        x, z = xz.chunk(2, dim=1)
        conv_state.copy_(x[:, :, -self.d_conv:])                                      # reference :315
        x = causal_conv1d_fn(x, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_dbl = self.x_proj(x.transpose(1, 2).reshape(batch * seqlen, self.d_inner))
        dt, Bm, Cm = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = (self.dt_proj.weight @ dt.t()).reshape(self.d_inner, batch, seqlen).transpose(0, 1)
        Bm = Bm.reshape(batch, seqlen, self.d_state).transpose(1, 2).contiguous()
        Cm = Cm.reshape(batch, seqlen, self.d_state).transpose(1, 2).contiguous()
        y, last_state = selective_scan_fn(x, dt, A, Bm, Cm, self.D.float(), z=z, delta_bias=self.dt_proj.bias.float(),
                                          delta_softplus=True, return_last_state=True)
        ssm_state.copy_(last_state)
        return self.out_proj(y.transpose(1, 2))

    def step(self, hidden_states, conv_state, ssm_state):
        """One token: hidden_states (B, 1, d_model); conv_state (B, d_inner, d_conv) and ssm_state (B, d_inner, d_state) are
        updated in place.  Reference :356-401."""
        from .causal_conv1d_interface import causal_conv1d_update
        from .selective_state_update import selective_state_update
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        xz = self.in_proj(hidden_states.squeeze(1))                                   # (B, 2*d_inner)
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x, conv_state, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_db = self.x_proj(x)
        dt, Bm, Cm = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = torch.nn.functional.linear(dt, self.dt_proj.weight)                      # the bias is added by the kernel
        A = -torch.exp(self.A_log.float())
        y = selective_state_update(ssm_state, x, dt, A, Bm, Cm, self.D, z=z, dt_bias=self.dt_proj.bias, dt_softplus=True)
        out = self.out_proj(y)
        return out.unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        """Reference :403-416."""
        device = self.out_proj.weight.device
        conv_dtype = self.conv1d.weight.dtype if dtype is None else dtype
        conv_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_conv, device=device, dtype=conv_dtype)
        ssm_dtype = self.dt_proj.weight.dtype if dtype is None else dtype
        ssm_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_state, device=device, dtype=ssm_dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        """Reference :418-445."""
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            conv_state, ssm_state = self.allocate_inference_cache(batch_size, 0)
            inference_params.key_value_memory_dict[self.layer_idx] = (conv_state, ssm_state)
        else:
            conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
            if initialize_states:
                conv_state.zero_()
                ssm_state.zero_()
        return conv_state, ssm_state
