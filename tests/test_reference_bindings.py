"""INTEGRATION.md section B, executed: the REFERENCE's own `mamba_ssm/ops/selective_scan_interface.py` running on top of
`segmamba_amd.native_stubs.{selective_scan_cuda, causal_conv1d_cuda}` (the two module names it imports, :9-11).

  * here (build container: /root/reference present, no GPU): the reference file is imported unchanged, the stubs are bound to
    the CPU emulation build of the kernels, and the reference's autograd Functions `selective_scan_fn` and
    `mamba_inner_fn_no_out_proj` are compared - output and every gradient - with the reference's own `selective_scan_ref` /
    `mamba_inner_ref`;
  * on the GPU box (-m gpu; no /root/reference there): the stubs are driven with the exact call sequences of those
    Functions (selective_scan_interface.py:36-67 and :213-247) on the HIP library and compared with the oracle.
"""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"


def _load_reference_interface():
    """the reference's selective_scan_interface module with ITS native imports resolved to our stubs"""
    from segmamba_amd import native_stubs
    native_stubs.install()
    saved = {k: sys.modules.get(k) for k in ("mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.selective_scan_interface",
                                              "causal_conv1d", "causal_conv1d.causal_conv1d_interface")}
    for k in saved:
        sys.modules.pop(k, None)
    pkg = types.ModuleType("mamba_ssm")
    pkg.__path__ = [os.path.join(REF, "mamba", "mamba_ssm")]       # skip mamba_ssm/__init__.py (LM imports)
    sys.modules["mamba_ssm"] = pkg
    sys.path.insert(0, os.path.join(REF, "causal-conv1d"))
    try:
        import mamba_ssm.ops.selective_scan_interface as ssi       # the reference's file, unchanged
    finally:
        sys.path.pop(0)
    return ssi, saved


def _restore(saved):
    for k, v in saved.items():
        sys.modules.pop(k, None)
        if v is not None:
            sys.modules[k] = v
    for k in ("selective_scan_cuda", "causal_conv1d_cuda"):
        sys.modules.pop(k, None)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
def test_reference_python_on_the_stub_modules(monkeypatch):
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    from segmamba_amd import lib as L
    monkeypatch.setattr(L, "_lib", emu_util.emu_lib())
    monkeypatch.setattr(L, "on_device", lambda t: True)
    ssi, saved = _load_reference_interface()
    try:
        assert ssi.__file__.startswith(REF)
        g = torch.Generator().manual_seed(0)
        # --- standalone op: reference test distributions (mamba/tests/ops/test_selective_scan.py:53-88), G = 1 and 2
        for G in (1, 2):
            B_, D_, N_, L_ = 2, 8, 16, 200
            A = (-0.5 * torch.rand(D_, N_, generator=g)).requires_grad_()
            shape = (B_, N_, L_) if G == 1 else (B_, G, N_, L_)
            Bm, Cm = (torch.randn(*shape, generator=g).requires_grad_() for _ in range(2))
            Dv = torch.randn(D_, generator=g).requires_grad_()
            z, u = (torch.randn(B_, D_, L_, generator=g).requires_grad_() for _ in range(2))
            db = (0.5 * torch.rand(D_, generator=g)).requires_grad_()
            delta = (0.5 * torch.rand(B_, D_, L_, generator=g)).requires_grad_()
            leaves = [u, delta, A, Bm, Cm, Dv, z, db]
            ref_leaves = [t.detach().clone().requires_grad_() for t in leaves]
            out, last = ssi.selective_scan_fn(*leaves, delta_softplus=True, return_last_state=True)
            out_ref, last_ref = ssi.selective_scan_ref(*ref_leaves, delta_softplus=True, return_last_state=True)
            assert torch.allclose(out, out_ref, rtol=1e-3, atol=1e-3)
            assert torch.allclose(last, last_ref, rtol=1e-3, atol=1e-3)
            gout = torch.randn(out.shape, generator=g)
            out.backward(gout)
            out_ref.backward(gout)
            for a, b, name in zip(leaves, ref_leaves, "u delta A B C D z delta_bias".split()):
                scale = max(1.0, float(b.grad.abs().max()))
                assert (a.grad - b.grad).abs().max() <= 1e-3 * scale, name
        # --- the fused function SegMamba uses (reference :155-289) against mamba_inner_ref without the out-projection
        B_, D_, N_, L_, R_ = 2, 16, 16, 96, 2
        xz = torch.randn(B_, 2 * D_, L_, generator=g).requires_grad_()
        conv_w = (0.5 * torch.randn(D_, 1, 4, generator=g)).requires_grad_()
        conv_b = (0.1 * torch.randn(D_, generator=g)).requires_grad_()
        xw = (0.3 * torch.randn(R_ + 2 * N_, D_, generator=g)).requires_grad_()
        dtw = (0.3 * torch.randn(D_, R_, generator=g)).requires_grad_()
        A = (-0.5 * torch.rand(D_, N_, generator=g)).requires_grad_()
        Dv = torch.randn(D_, generator=g).requires_grad_()
        db = (0.5 * torch.rand(D_, generator=g)).requires_grad_()
        leaves = [xz, conv_w, conv_b, xw, dtw, A, Dv, db]
        ref_leaves = [t.detach().clone().requires_grad_() for t in leaves]
        y = ssi.mamba_inner_fn_no_out_proj(xz, conv_w, conv_b, xw, dtw, A, None, None, Dv, db, None, None, True)
        import causal_conv1d.causal_conv1d_interface as cci
        monkeypatch.setattr(ssi, "causal_conv1d_fn", cci.causal_conv1d_ref)
        r = ref_leaves
        y_ref = ssi.mamba_inner_ref(r[0], r[1], r[2], r[3], r[4], torch.eye(D_), None, r[5], None, None, r[6], r[7], None, None,
                                    True).transpose(1, 2)
        assert torch.allclose(y, y_ref, rtol=1e-3, atol=1e-3)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        y_ref.backward(gy)
        for a, b, name in zip(leaves, ref_leaves, "xz conv_w conv_b x_proj_w dt_proj_w A D delta_bias".split()):
            scale = max(1.0, float(b.grad.abs().max()))
            assert (a.grad - b.grad).abs().max() <= 1e-3 * scale, name
    finally:
        _restore(saved)


@pytest.mark.gpu
def test_stub_modules_with_the_reference_call_sequences_on_the_gpu():
    """what SelectiveScanFn.forward / backward and MambaInnerFnNoOutProj do with the native modules, on the HIP library"""
    from oracle import ref_ops
    from segmamba_amd.native_stubs import causal_conv1d_cuda, selective_scan_cuda
    from tests import helpers as H
    dev = "cuda"
    c = H.scan_case(2, 96, 16, 1536, seed=5)
    ref = H.scan_oracle(c)
    t = {k: (v.to(dev) if v is not None else None) for k, v in c.items()}
    B4, C4 = t["B"][:, None].contiguous(), t["C"][:, None].contiguous()            # "b dstate l -> b 1 dstate l" (:31-36)
    out, x, out_z = selective_scan_cuda.fwd(t["u"], t["delta"], t["A"], B4, C4, t["D"], t["z"], t["delta_bias"], True)
    assert isinstance(x, torch.Tensor)                                             # ctx.save_for_backward(..., x, out)
    last = x[:, :, -1, 1::2]                                                       # :40
    H.assert_close(out_z, ref["out"], 1e-3, 1e-3, "stub fwd out_z")
    H.assert_close(last, ref["last_state"], 1e-3, 1e-3, "stub fwd last_state")
    x_saved = x.detach()[...]                                                      # what autograd hands back: same storage
    du, ddelta, dA, dB, dC, dD, ddb, dz = selective_scan_cuda.bwd(t["u"], t["delta"], t["A"], B4, C4, t["D"], t["z"],
                                                                    t["delta_bias"], t["g"], x_saved, out, None, True, False)
    res = {"out": out_z, "last_state": last, "du": du, "ddelta": ddelta, "dA": dA, "dB": dB.squeeze(1), "dC": dC.squeeze(1),
           "dD": dD, "dz": dz, "ddelta_bias": ddb}
    H.check_scan(res, ref, torch.float32, "stub modules")
    # dz pre-allocated as a view into a (B, 2D, L) buffer + recompute_out_z (:244-247)
    dxz = torch.empty(2, 192, 1536, device=dev)
    r2 = selective_scan_cuda.bwd(t["u"], t["delta"], t["A"], B4, C4, t["D"], t["z"], t["delta_bias"], t["g"], x_saved, out,
                                 dxz[:, 96:], True, True)
    assert len(r2) == 9 and r2[7].data_ptr() == dxz[:, 96:].data_ptr()
    H.assert_close(dxz[:, 96:], ref["dz"], 1e-3, 1e-3, "dz in place")
    H.assert_close(r2[8], ref["out"], 1e-3, 1e-3, "recomputed out_z")
    # conv module: fwd, bwd with pre-allocated dx, 16-bit weights (causal_conv1d.cpp:136-137)
    g = torch.Generator().manual_seed(3)
    xx = torch.randn(2, 64, 300, generator=g)
    w, b = torch.randn(64, 4, generator=g), torch.randn(64, generator=g)
    go = torch.randn(2, 64, 300, generator=g)
    xr, wr, br = xx.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = ref_ops.causal_conv1d_ref(xr, wr, br, "silu")
    yr.backward(go)
    y = causal_conv1d_cuda.causal_conv1d_fwd(xx.to(dev), w.to(dev), b.to(dev), True)
    H.assert_close(y, yr, 1e-3, 1e-3, "conv stub fwd")
    dxbuf = torch.empty(2, 128, 300, device=dev)
    dx, dw, dbias = causal_conv1d_cuda.causal_conv1d_bwd(xx.to(dev), w.to(dev), b.to(dev), go.to(dev), dxbuf[:, :64], True)
    assert dx.data_ptr() == dxbuf.data_ptr()
    H.assert_close(dx, xr.grad, 1e-3, 1e-3, "conv stub dx")
    H.assert_close(dw, wr.grad, 1e-3, 1e-3 * float(wr.grad.abs().max()), "conv stub dw")
    H.assert_close(dbias, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()), "conv stub db")
    y16 = causal_conv1d_cuda.causal_conv1d_fwd(xx.to(dev).bfloat16(), w.to(dev).bfloat16(), b.to(dev).bfloat16(), True)
    yr16 = ref_ops.causal_conv1d_ref(xx.bfloat16(), w.bfloat16().float(), b.bfloat16().float(), "silu")
    H.assert_close(y16, yr16, 1e-2, 1e-2, "conv stub fwd, bf16 weights")
    with pytest.raises(RuntimeError):
        selective_scan_cuda.fwd(t["u"], t["delta"], torch.randn(96, 32, device=dev), B4, C4, t["D"], t["z"], t["delta_bias"], True)
