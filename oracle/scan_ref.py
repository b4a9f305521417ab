"""CPU ORACLE (test infrastructure, never a product path): ctypes front end of oracle/scan_ref.c.

fp64 selective scan forward / backward and causal conv1d on the host cores (OpenMP), for the sizes BASELINE.json names
(L = 64^3 ... 2^24) that the PyTorch oracle (ref_ops.py, a Python loop over time) cannot walk in test time.  Same
layouts as the reference's `selective_scan_ref` (mamba/mamba_ssm/ops/selective_scan_interface.py:86-152): u, delta, z
(B, D, L); A (D, N); B, C (B, N, L) or (B, G, N, L).  Inputs are converted to fp32 (exact for fp32 / bf16 / fp16 test
inputs), all arithmetic is double, results come back as float64 torch tensors.

Pinned against the reference-generated fixtures by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import build_oracle

_dll = None


def lib():
    global _dll
    if _dll is None:
        _dll = C.CDLL(build_oracle.build())
        _dll.segm_oracle_threads.restype = C.c_int
        fp, dp = C.c_void_p, C.c_void_p
        _dll.segm_oracle_scan_fwd.argtypes = [C.c_int] * 4 + [C.c_int64] + [fp] * 8 + [C.c_int] + [dp] * 3
        _dll.segm_oracle_scan_bwd.argtypes = [C.c_int] * 4 + [C.c_int64] + [fp] * 8 + [C.c_int, fp] + [dp] * 8
        _dll.segm_oracle_conv1d_fwd.argtypes = [C.c_int] * 3 + [C.c_int64, fp, fp, fp, C.c_int, dp]
    return _dll


def threads() -> int:
    return int(lib().segm_oracle_threads())


def _f32(t):
    return None if t is None else np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())


def _p(a):
    return None if a is None else a.ctypes.data


def _bc(t):
    a = _f32(t)
    return a[:, None] if a.ndim == 3 else a


def scan_fwd(u, delta, A, B, C_, D=None, z=None, delta_bias=None, delta_softplus=False, want_y=False):
    """-> dict(out (B, D, L) float64 [gated by silu(z) when z is given], last_state (B, D, N), [y])"""
    u_, dl, A_, Bm, Cm, D_, z_, db = _f32(u), _f32(delta), _f32(A), _bc(B), _bc(C_), _f32(D), _f32(z), _f32(delta_bias)
    Bn, Dn, Ln = u_.shape
    N, G = A_.shape[1], Bm.shape[1]
    Bm, Cm = np.ascontiguousarray(Bm), np.ascontiguousarray(Cm)
    out = np.empty((Bn, Dn, Ln), np.float64)
    y = np.empty((Bn, Dn, Ln), np.float64) if want_y else None
    last = np.empty((Bn, Dn, N), np.float64)
    rc = lib().segm_oracle_scan_fwd(Bn, Dn, N, G, Ln, _p(u_), _p(dl), _p(A_), _p(Bm), _p(Cm), _p(D_), _p(z_), _p(db),
                                    int(bool(delta_softplus)), _p(out), _p(y), _p(last))
    assert rc == 0
    r = {"out": torch.from_numpy(out), "last_state": torch.from_numpy(last)}
    if want_y:
        r["y"] = torch.from_numpy(y)
    return r


def scan_bwd(u, delta, A, B, C_, D, z, delta_bias, dout, delta_softplus=False, want_bc=True):
    """All eight gradients of sum(out * dout), float64, in the reference's layouts (dB / dC shaped like B / C)."""
    u_, dl, A_, Bm, Cm, D_, z_, db, g = (_f32(u), _f32(delta), _f32(A), _bc(B), _bc(C_), _f32(D), _f32(z), _f32(delta_bias),
                                        _f32(dout))
    Bn, Dn, Ln = u_.shape
    N, G = A_.shape[1], Bm.shape[1]
    Bm, Cm = np.ascontiguousarray(Bm), np.ascontiguousarray(Cm)
    du, dd = np.empty((Bn, Dn, Ln)), np.empty((Bn, Dn, Ln))
    dz = np.empty((Bn, Dn, Ln)) if z is not None else None
    dA, dD, ddb = np.empty((Dn, N)), np.empty(Dn), np.empty(Dn)
    dB = np.empty((Bn, G, N, Ln)) if want_bc else None
    dC = np.empty((Bn, G, N, Ln)) if want_bc else None
    rc = lib().segm_oracle_scan_bwd(Bn, Dn, N, G, Ln, _p(u_), _p(dl), _p(A_), _p(Bm), _p(Cm), _p(D_), _p(z_), _p(db),
                                    int(bool(delta_softplus)), _p(g), _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz),
                                    _p(ddb))
    assert rc == 0
    r = {"du": torch.from_numpy(du), "ddelta": torch.from_numpy(dd), "dA": torch.from_numpy(dA)}
    if want_bc:
        squeeze = B.dim() == 3
        r["dB"] = torch.from_numpy(dB[:, 0] if squeeze else dB)
        r["dC"] = torch.from_numpy(dC[:, 0] if squeeze else dC)
    if D is not None:
        r["dD"] = torch.from_numpy(dD)
    if z is not None:
        r["dz"] = torch.from_numpy(dz)
    if delta_bias is not None:
        r["ddelta_bias"] = torch.from_numpy(ddb)
    return r


def conv1d_fwd(x, weight, bias=None, silu=True):
    x_, w_, b_ = _f32(x), _f32(weight.reshape(weight.shape[0], -1)), _f32(bias)
    Bn, Dn, Ln = x_.shape
    out = np.empty((Bn, Dn, Ln), np.float64)
    rc = lib().segm_oracle_conv1d_fwd(Bn, Dn, w_.shape[1], Ln, _p(x_), _p(w_), _p(b_), int(bool(silu)), _p(out))
    assert rc == 0
    return torch.from_numpy(out)
