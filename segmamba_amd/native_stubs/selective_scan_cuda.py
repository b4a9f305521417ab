"""`selective_scan_cuda` - stands in for the pybind module of mamba/csrc/selective_scan/selective_scan.cpp:494-497.

    fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus) -> [out, x, (out_z)]                 selective_scan.cpp:226-336
    bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute_out_z)
        -> [du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)]                                selective_scan.cpp:338-492

Layouts are the reference's: u, delta, z, dout (B, D, L) with stride(-1) == 1; A (D, N) fp32; B, C (B, G, N, L); D,
delta_bias (D) fp32.  `x` - the reference's per-chunk scan state (B, D, n_chunks, 2N) - is opaque to callers except for
`x[:, :, -1, 1::2]`, the last state (selective_scan_interface.py:40), and must be a tensor (it goes through
ctx.save_for_backward).  Here it is a (B, D, 1, 2N) fp32 view at the end of one buffer whose head holds this library's
state checkpoints; `bwd` finds them again through the view's storage.
"""
import torch

from .. import lib as L
from .. import ops_raw


def _ckpt_floats(hip, batch, dim, dstate, seqlen):
    return hip.dll.segm_selective_scan_ckpt_bytes(batch, dim, dstate, seqlen) // 4


def _check(A):
    if A.is_complex():
        raise RuntimeError("selective_scan_cuda (MI355X): complex A is not supported (not on the SegMamba path)")
    if A.shape[-1] > 16:
        raise RuntimeError("selective_scan_cuda (MI355X): dstate > 16 at this level - call the drop-in "
                           "mamba_ssm.ops.selective_scan_interface.selective_scan_fn, which splits wider states into blocks")


def fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    _check(A)
    if B.dim() < 3 or C.dim() < 3:
        raise RuntimeError("selective_scan_cuda (MI355X): B and C must be input-dependent, (B, G, N, L)")
    hip = L.get_lib()
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n = _ckpt_floats(hip, batch, dim, dstate, seqlen)
    buf = torch.empty(n + batch * dim * 2 * dstate, dtype=torch.float32, device=u.device)
    r = ops_raw.scan_fwd(hip, u, delta, A.float().contiguous(), B, C, None if D is None else D.float(), z,
                         None if delta_bias is None else delta_bias.float(), bool(delta_softplus), channel_last=False,
                         need_out=True, need_ckpt=True, need_last_state=True, ckpt_buf=buf[:n])
    x = buf[n:].view(batch, dim, 1, 2 * dstate)
    x[:, :, 0, 0::2] = 1.0                                 # the reference keeps the running decay product here; unused by callers
    x[:, :, 0, 1::2] = r["last_state"]
    return [r["out"], x, r["out_z"]] if z is not None else [r["out"], x]


def bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute_out_z):
    _check(A)
    hip = L.get_lib()
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n = _ckpt_floats(hip, batch, dim, dstate, seqlen)
    whole = torch.empty(0, dtype=torch.float32, device=x.device).set_(x.untyped_storage())
    if whole.numel() < n + batch * dim * 2 * dstate:
        raise RuntimeError("selective_scan_cuda.bwd: `x` is not the tensor fwd returned")
    ckpt = whole[:n]
    chunk = hip.dll.segm_selective_scan_default_chunk(batch, dim, seqlen)      # what fwd used
    if z is not None and out is None:
        raise RuntimeError("selective_scan_cuda.bwd: `out` is required when z is given")          # selective_scan.cpp:430-433
    g = ops_raw.scan_bwd(hip, u, delta, A.float().contiguous(), B, C, None if D is None else D.float(), z,
                         None if delta_bias is None else delta_bias.float(), dout, out, ckpt, bool(delta_softplus),
                         channel_last=False, chunk=chunk, dz=dz)                 # dz may be a view into dxz (:244-245)
    # reference: dB / dC are accumulated in fp32 and cast to the input dtype on return (selective_scan.cpp:461-462,488)
    res = [g["du"], g["ddelta"], g["dA"], g["dB"].to(B.dtype), g["dC"].to(C.dtype),
           g["dD"] if D is not None else torch.zeros(dim, dtype=torch.float32, device=u.device),
           g["ddelta_bias"] if delta_bias is not None else torch.zeros(dim, dtype=torch.float32, device=u.device)]
    if z is not None:
        res.append(g["dz"])
        if recompute_out_z:                                                      # :247 asks for it and drops it
            res.append((out.float() * torch.nn.functional.silu(z.float())).to(out.dtype))
    return res
