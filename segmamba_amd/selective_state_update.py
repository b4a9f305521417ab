"""`selective_state_update` (single-token SSM step) on the MI355X library.

Host-side mirror of reference mamba/mamba_ssm/ops/triton/selective_state_update.py:99-155 (a Triton kernel there,
`segm_selective_state_update` here): same name, argument meaning and in-place update of `state`.
"""
from __future__ import annotations

from . import lib as L
from . import ops_raw


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """
    state: (batch, dim, dstate), updated in place      x, dt, z: (batch, dim)      A: (dim, dstate)
    B, C: (batch, dstate)      D, dt_bias: (dim,)
    out: (batch, dim)
    """
    batch, dim, dstate = state.shape
    assert x.shape == (batch, dim)
    assert dt.shape == x.shape
    assert A.shape == (dim, dstate)
    assert B.shape == (batch, dstate)
    assert C.shape == B.shape
    if D is not None:
        assert D.shape == (dim,)
    if z is not None:
        assert z.shape == x.shape
    if dt_bias is not None:
        assert dt_bias.shape == (dim,)
    f32 = lambda t: None if t is None else t.float().contiguous()                      # noqa: E731
    cast = lambda t: None if t is None else t.to(x.dtype)                              # noqa: E731
    return ops_raw.state_update(L.get_lib(), state, x, cast(dt), f32(A), cast(B), cast(C), f32(D), cast(z), f32(dt_bias),
                                dt_softplus)
