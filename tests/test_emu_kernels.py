"""CPU checks of the KERNEL SOURCES themselves: segmamba_amd/csrc/*.hip compiled unchanged against the small CPU
emulation of the HIP runtime in tests/emu (OS threads for HIP threads, barriers for __syncthreads, ...) and driven
through the same C ABI + host marshalling as the GPU build.  This is test infrastructure (it lets index arithmetic,
masking, chunk/carry composition and the time-order maps be validated in the GPU-less build container); parity on the
real device is tests/test_gpu_*.py."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests import emu_util
from segmamba_amd import lib as L
from segmamba_amd import ops_raw
from oracle import ref_ops

pytestmark = pytest.mark.skipif(not emu_util.emu_available(), reason="ROCm host clang not present")
# the fp32 references of this file are torch CPU convolutions: oneDNN's weight gradient of strided 3-D convolutions returns
# garbage for some shapes on this PyTorch build (tools/emu_random_sweep_conv.py found one), the native path does not
torch.backends.mkldnn.enabled = False


@pytest.fixture(scope="module")
def emu():
    return emu_util.emu_lib()


@pytest.mark.parametrize("dim,dstate,seqlen,chunk,channel_last,order,ns,groups", [
    (4, 8, 100, 32, True, L.TIME_FORWARD, 1, 1),          # reference test dims, ragged tail
    (96, 16, 80, 32, True, L.TIME_FORWARD, 1, 1),         # SegMamba stage-0 width: 32-channel work items, 2 per wave
    (40, 16, 70, 64, False, L.TIME_FORWARD, 1, 1),        # channel-first (reference layout), padded d-tile
    (64, 16, 96, 32, True, L.TIME_REVERSED, 1, 1),
    (32, 16, 96, 32, True, L.TIME_INTERLEAVED, 8, 1),
    (8, 8, 64, 32, True, L.TIME_FORWARD, 1, 2),           # grouped B / C
    (64, 4, 64, 32, True, L.TIME_FORWARD, 1, 1),          # 4 states: a work item is narrowed to the 8 x 4 staging block
    (64, 3, 48, 16, False, L.TIME_REVERSED, 1, 1),
])
def test_scan_forward_backward_emulated(emu, dim, dstate, seqlen, chunk, channel_last, order, ns, groups):
    c = H.scan_case(2 if dim <= 8 else 1, dim, dstate, seqlen, groups=groups, seed=dim + seqlen)
    ref = H.scan_oracle(c, order, ns)
    res = H.run_scan(emu, c, "cpu", channel_last, order, ns, chunk=chunk)
    H.check_scan(res, ref, torch.float32, f"emu D={dim} L={seqlen}")


@pytest.mark.parametrize("seqlen,order,ns", [(16 * 70, L.TIME_FORWARD, 1), (16 * 130, L.TIME_REVERSED, 1), (16 * 72, L.TIME_INTERLEAVED, 8)])
def test_scan_many_chunks_carry_segments_emulated(emu, seqlen, order, ns):
    """more chunks than one carry segment holds (64): the two-launch carry composition (segment composites, then the walk
    from the composed state), forward and - in the backward pass - reversed, with a partly filled last segment"""
    c = H.scan_case(1, 16, 16, seqlen, seed=seqlen)
    ref = H.scan_oracle(c, order, ns)
    res = H.run_scan(emu, c, "cpu", True, order, ns, chunk=16)
    H.check_scan(res, ref, torch.float32, f"emu many chunks L={seqlen}")


def test_scan_bf16_and_no_gate_emulated(emu):
    c = H.scan_case(1, 32, 16, 64, dtype=torch.bfloat16)
    H.check_scan(H.run_scan(emu, c, "cpu", True, chunk=32), H.scan_oracle(c), torch.bfloat16, "emu bf16")
    c = H.scan_case(1, 16, 16, 40, has_z=False, has_D=False, has_bias=False)
    H.check_scan(H.run_scan(emu, c, "cpu", True, chunk=32, softplus=False), H.scan_oracle(c, softplus=False),
                 torch.float32, "emu plain")


@pytest.mark.parametrize("width,order,ns,channel_last", [(4, L.TIME_FORWARD, 1, True), (3, L.TIME_FORWARD, 1, False),
                                                         (2, L.TIME_REVERSED, 1, True), (4, L.TIME_INTERLEAVED, 8, True)])
def test_conv1d_emulated(emu, width, order, ns, channel_last):
    torch.manual_seed(width)
    b, dim, seqlen = 2, 24, 152
    x, w, bias, g = torch.randn(b, dim, seqlen), torch.randn(dim, width), torch.randn(dim), torch.randn(b, dim, seqlen)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), bias.clone().requires_grad_()
    ref = H.iperm(ref_ops.causal_conv1d_ref(H.perm(xr, order, ns), wr, br, "silu"), order, ns)
    ref.backward(g)
    tr = (lambda t: t.transpose(1, 2).contiguous()) if channel_last else (lambda t: t)
    out = ops_raw.conv1d_fwd(emu, tr(x), w, bias, True, channel_last=channel_last, time_order=order, nslices=ns)
    dx, dw, db = ops_raw.conv1d_bwd(emu, tr(x), w, bias, tr(g), True, channel_last=channel_last, time_order=order, nslices=ns)
    back = (lambda t: t.transpose(1, 2)) if channel_last else (lambda t: t)
    H.assert_close(back(out), ref, 3e-4, 1e-3, "out")
    H.assert_close(back(dx), xr.grad, 3e-4, 1e-3, "dx")
    H.assert_close(dw, wr.grad, 1e-3, 1e-3, "dweight")
    H.assert_close(db, br.grad, 1e-3, 1e-3, "dbias")


def test_conv1d_golden_from_reference_emulated(emu):
    f = H.load_golden("conv1d_w4.npz")
    out = ops_raw.conv1d_fwd(emu, f["x"], f["weight"], f["bias"], True)
    dx, dw, db = ops_raw.conv1d_bwd(emu, f["x"], f["weight"], f["bias"], f["g"], True)
    H.assert_close(out, f["out"], 3e-4, 1e-3, "out")
    H.assert_close(dx, f["dx"], 3e-4, 1e-3, "dx")
    H.assert_close(dw, f["dweight"], 1e-3, 1e-3, "dweight")
    H.assert_close(db, f["dbias"], 1e-3, 1e-3, "dbias")


def test_scan_golden_from_reference_emulated(emu):
    f = H.load_golden("scan_L64_G2.npz")
    c = {k: f[k] for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "g")}
    ref = {k: f[k] for k in ("out", "last_state", "du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias")}
    H.check_scan(H.run_scan(emu, c, "cpu", False, chunk=32), ref, torch.float32, "golden")


def test_host_autograd_and_mamba_v3_on_emulated_kernels(emu, monkeypatch):
    """The product's host code (autograd Functions, Mamba module) end to end against the reference's Mamba(v3) fixture,
    with the kernels emulated.  Patching `lib._lib` is something only a test does."""
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm import Mamba
    from tests.golden.make_golden import named_fill
    f = H.load_golden("mamba_v3.npz")
    m = Mamba(d_model=16, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=int(f["nslices"]))
    m.load_state_dict(named_fill(m.state_dict()))
    x = f["x"].clone().requires_grad_()
    y = m(x)
    y.backward(f["g"])
    H.assert_close(y, f["y"], 1e-4, 1e-5, "y")
    H.assert_close(x.grad, f["dx"], 1e-4, 1e-5, "dx")
    for k, p in m.named_parameters():
        r = f["grad__" + k]
        H.assert_close(p.grad, r, 1e-3, 1e-3 * max(1e-3, float(r.abs().max())), "grad " + k)


def test_reference_layout_inner_fn_on_emulated_kernels(emu, monkeypatch):
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn_no_out_proj
    f = H.load_golden("inner_no_out_proj.npz")
    names = ("xz", "conv_w", "conv_b", "x_proj_w", "dt_proj_w", "A", "D", "delta_bias")
    t = {k: f[k].clone().requires_grad_() for k in names}
    out = mamba_inner_fn_no_out_proj(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["A"], None, None,
                                     t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    out.backward(f["g"])
    H.assert_close(out, f["out"], 1e-3, 1e-4, "out")
    for k, gk in (("xz", "dxz"), ("conv_w", "dconv_w"), ("conv_b", "dconv_b"), ("x_proj_w", "dx_proj_w"),
                  ("dt_proj_w", "ddt_proj_w"), ("A", "dA"), ("D", "dD"), ("delta_bias", "ddelta_bias")):
        H.assert_close(t[k].grad, f[gk], 1e-3, 1e-3 * max(1.0, float(f[gk].abs().max())), gk)


@pytest.mark.parametrize("vB,vC", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_mamba_inner_fn_matrix_golden_on_emulated_kernels(emu, monkeypatch, vB, vC):
    """`mamba_inner_fn` (with the output projection) against the reference's `mamba_inner_ref` over the reference test's
    matrix of input-dependent / constant B and C (mamba/tests/ops/test_selective_scan.py:152-221; fixtures from
    tests/golden/make_golden_inner_out_proj.py): output and EVERY gradient"""
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    f = H.load_golden(f"inner_fn_vB{vB}_vC{vC}.npz")
    out, grads = H.run_inner_fn(f, "cpu")
    H.check_inner_fn(out, grads, f, f"emu inner_fn vB{vB} vC{vC}")


def test_bimamba_inner_fn_golden_on_emulated_kernels(emu, monkeypatch):
    """`bimamba_inner_fn` against the reference's `bimamba_inner_ref` (selective_scan_interface.py:673-709)"""
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    f = H.load_golden("bimamba_inner.npz")
    out, grads = H.run_inner_fn(f, "cpu", bidirectional=True)
    H.check_inner_fn(out, grads, f, "emu bimamba_inner_fn")


def _wgrad_reference(x, dy):
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, 3, requires_grad=True)
    torch.nn.functional.conv3d(x.float(), w, None, 1, 1).backward(dy.float())
    return w.grad


@pytest.mark.parametrize("shape", [(1, 48, 48, 3, 4, 32), (1, 96, 48, 2, 3, 64), (1, 48, 48, 1, 20, 64), (1, 48, 48, 2, 5, 16), (1, 48, 48, 2, 3, 40), (1, 48, 48, 1, 3, 128)])
def test_conv3d_k3_wgrad_emulated(emu, shape):
    """MFMA weight-gradient kernel (fragment layout, halo / funnel-shift x taps, z / y border masking, slab reduce)."""
    B, cin, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, cin, D, H_, W, generator=g).bfloat16()
    dy = torch.randn(B, cout, D, H_, W, generator=g).bfloat16()
    ref = _wgrad_reference(x, dy)
    dw = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.float32)
    assert (dw - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-4
    dwb = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.bfloat16)
    assert torch.equal(dwb, dw.bfloat16())


def test_conv3d_k3_wgrad_channel_slices_and_errors_emulated(emu):
    g = torch.Generator().manual_seed(5)
    xb = torch.randn(1, 96, 2, 2, 32, generator=g).bfloat16()
    dyb = torch.randn(1, 96, 2, 2, 32, generator=g).bfloat16()
    xs, dys = xb[:, 48:], dyb[:, :48]                       # views: the cat-free decoder passes channel slices
    ref = _wgrad_reference(xs, dys)
    dw = ops_raw.conv3d_k3_wgrad(emu, xs, dys, torch.float32)
    assert (dw - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-4
    assert not ops_raw.conv3d_k3_wgrad_supported(xb[:, :56], dyb[:, :48])          # cin > 48 and not a multiple of 48
    assert not ops_raw.conv3d_k3_wgrad_supported(xb[..., :12], dyb[..., :12])      # width % 8
    assert not ops_raw.conv3d_k3_wgrad_supported(xs.float(), dys.float())          # dtype
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_k3_wgrad(emu, xb[:, :56], dyb[:, :48])


def _instnorm_reference(x, res, act, slope, gy):
    import torch.nn.functional as F
    x = x.double().requires_grad_()
    res = res.double().requires_grad_() if res is not None else None
    y = F.instance_norm(x, eps=1e-5)
    if res is not None:
        y = y + res
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky_relu":
        y = F.leaky_relu(y, slope)
    grads = torch.autograd.grad(y, (x,) if res is None else (x, res), gy.double())
    return y, grads


@pytest.mark.parametrize("shape,act,with_res,dtype", [
    ((2, 3, 4, 8, 16), "leaky_relu", False, torch.float32),      # vector path, one slab
    ((1, 2, 3, 5, 7), "relu", True, torch.float32),              # odd size: scalar path
    ((1, 2, 16, 32, 48), "leaky_relu", True, torch.float32),     # several slabs per instance (Chan merge)
    ((2, 2, 4, 8, 16), "none", False, torch.bfloat16),
    ((1, 3, 4, 8, 16), "leaky_relu", True, torch.bfloat16),
    ((1, 2, 18, 32, 48), "leaky_relu", True, torch.bfloat16),    # 13.5 thread-strides per instance: the unrolled loops and their tails
    ((1, 2, 18, 32, 48), "relu", False, torch.bfloat16),
])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_instance_norm_act_emulated(emu, shape, act, with_res, dtype, mode, monkeypatch):
    """mode: how the passes stream (csrc/instnorm.hip norm_mode: the library picks by tensor size - 1 and 2, non-temporal accesses and
    four packets in flight per thread, are what the 64^3 / 128^3 levels take; forced here on test-sized tensors)"""
    monkeypatch.setenv("SEGM_NORM_NT", str(mode))
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dtype)
    res = torch.randn(shape, generator=g).to(dtype) if with_res else None
    gy = torch.randn(shape, generator=g).to(dtype)
    ref_y, ref_g = _instnorm_reference(x, res, act, 0.01, gy)
    y, mean, rstd = ops_raw.instnorm_fwd(emu, x, res, act, 0.01, 1e-5)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert (y.double() - ref_y).abs().max() < tol
    xd = x.double().flatten(2)
    assert torch.allclose(mean.double(), xd.mean(-1).flatten(), atol=1e-5)
    assert torch.allclose(rstd.double(), (xd.var(-1, unbiased=False) + 1e-5).rsqrt().flatten(), rtol=1e-4)
    dx, dres = ops_raw.instnorm_bwd(emu, x, gy, mean, rstd, y if (with_res and act != "none") else None, act, 0.01,
                                    want_dresidual=with_res)
    # the activation mask of the oracle is computed in fp64, the kernel's from the dtype-rounded values: compare away
    # from the kink
    assert (dx.double() - ref_g[0]).abs().max() < tol * 4
    if with_res:
        assert (dres.double() - ref_g[1]).abs().max() < tol


def _channel_padded(t, pad):
    """the same values in a buffer whose channel stride is padded by `pad` elements (what ops_raw.volume_empty hands out)"""
    B, C = t.shape[:2]
    S = t[0, 0].numel()
    buf = torch.full((B, C, S + pad), float("nan"), dtype=t.dtype)
    v = buf[:, :, :S].view(t.shape)
    v.copy_(t)
    return v


@pytest.mark.parametrize("shape,act,with_res,dtype,padx,pady", [
    ((2, 3, 4, 8, 16), "leaky_relu", True, torch.bfloat16, 192, 64),     # vector path on both padded operands
    ((1, 2, 16, 32, 48), "relu", False, torch.float32, 64, 0),           # several slabs, dy dense
    ((2, 2, 3, 5, 7), "leaky_relu", True, torch.float32, 3, 5),          # odd strides: the scalar path
])
def test_instance_norm_padded_channel_stride_emulated(emu, shape, act, with_res, dtype, padx, pady):
    """x, the residual, the saved y and dy with (different) padded channel strides - the volumes of the 128^3 level - give
    bit-identical results to the dense tensors; the padding is never read (it holds NaN)"""
    g = torch.Generator().manual_seed(sum(shape) + padx)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dtype)
    res = torch.randn(shape, generator=g).to(dtype) if with_res else None
    gy = torch.randn(shape, generator=g).to(dtype)
    y0, mean0, rstd0 = ops_raw.instnorm_fwd(emu, x, res, act, 0.01, 1e-5)
    ym = y0 if (with_res and act != "none") else None
    dx0, dres0 = ops_raw.instnorm_bwd(emu, x, gy, mean0, rstd0, ym, act, 0.01, want_dresidual=with_res)
    xp, gp = _channel_padded(x, padx), (_channel_padded(gy, pady) if pady else gy)
    rp = _channel_padded(res, pady + 8) if with_res else None
    assert ops_raw.channel_dense(xp) and not xp.is_contiguous()
    y1, mean1, rstd1 = ops_raw.instnorm_fwd(emu, xp, rp, act, 0.01, 1e-5)
    assert torch.equal(y1, y0) and torch.equal(mean1, mean0) and torch.equal(rstd1, rstd0)
    ymp = _channel_padded(ym, padx + 16) if ym is not None else None
    dx1, dres1 = ops_raw.instnorm_bwd(emu, xp, gp, mean1, rstd1, ymp, act, 0.01, want_dresidual=with_res)
    assert torch.equal(dx1, dx0)
    if with_res:
        assert torch.equal(dres1, dres0)


@pytest.mark.parametrize("kw", [dict(chain=True, pitch48=True), dict(chain32=True)])
@pytest.mark.parametrize("accumulate,stats", [(False, False), (True, False), (False, True), (True, True)])
def test_conv3d_wide_epilogue_is_the_narrow_one_emulated(emu, monkeypatch, kw, accumulate, stats):
    """the chained kernels' 16-byte epilogue (v_permlane16_swap_b32 pairs the wave's two x tiles; csrc/conv3d_fwd.hip store_pair)
    against their 8-byte epilogue (SEGM_CONV_WIDE=0): outputs and statistics partials bit for bit, plain / accumulate / with the
    statistics epilogue, on a padded destination"""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 48, 3, 5, 64, generator=g).bfloat16()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
    wp = ops_raw.pack_conv3d_weight(w)
    y0 = torch.randn(2, 48, 3, 5, 64, generator=g).bfloat16()
    outs = []
    for wide in ("1", "0"):
        monkeypatch.setenv("SEGM_CONV_WIDE", wide)
        out = _channel_padded(y0.clone(), 64)
        r = ops_raw.conv3d_k3_fwd(emu, x, wp, None, out=out, accumulate=accumulate, want_stats=stats, **kw)
        outs.append(r if stats else (r, None))
    (ya, sa), (yb, sb) = outs
    assert torch.equal(ya, yb)
    if stats:
        assert sa is not None and torch.equal(sa, sb)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1) + (y0.float() if accumulate else 0)
    assert (ya.float() - ref).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))


def test_conv3d_chain_padded_channel_stride_emulated(emu):
    """the 3x3x3 kernels (all variants) and the 1x1x1 kernel on input / output volumes with a padded channel stride"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 48, 3, 6, 64, generator=g).bfloat16()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
    wp = ops_raw.pack_conv3d_weight(w)
    xp = _channel_padded(x, 192)
    for kw in ({}, dict(chain=True), dict(chain=True, pitch48=True), dict(chain32=True)):
        ref = ops_raw.conv3d_k3_fwd(emu, x, wp, None, **kw)
        out = _channel_padded(torch.zeros_like(ref), 64)
        ops_raw.conv3d_k3_fwd(emu, xp, wp, None, out=out, **kw)
        assert torch.equal(out, ref), kw
    w2 = (0.1 * torch.randn(32, 48, generator=g)).bfloat16()
    ref = ops_raw.pointwise_cf(emu, x.flatten(2), w2, None)
    out = _channel_padded(torch.zeros_like(ref), 64)
    ops_raw.pointwise_cf(emu, xp.flatten(2), w2, None, out=out)
    assert torch.equal(out, ref)


def test_unet_res_block_on_padded_volumes_emulated(emu, monkeypatch):
    """A UnetResBlock (3x3x3 conv -> IN -> LeakyReLU -> 3x3x3 conv -> IN, 1x1x1 conv -> IN skip, add, LeakyReLU) forward and
    backward through the library with every volume allocated by volume_empty PADDED (quantum lowered so that the small test
    volumes qualify, as the 128^3 level does at full size) against the same block on dense volumes: bit-identical"""
    from segmamba_amd import lib as L, unet_blocks as UB, conv3d as C3
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())     # the library's candidates (forward, dgrad, wgrad)
    torch.manual_seed(3)
    blk = UB.UnetResBlock(96, 48).bfloat16()
    g = torch.Generator().manual_seed(9)
    xa = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    xb = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    dy = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    res = []
    for quantum in (1 << 20, 1024):                       # 2 * 4 * 64 * 2 B = 1024-byte channels
        monkeypatch.setattr(ops_raw, "_VOLUME_PAD_QUANTUM", quantum)
        a, b = xa.clone().requires_grad_(), xb.clone().requires_grad_()
        blk.zero_grad()
        y = blk((a, b))
        assert y.is_contiguous() == (quantum != 1024)
        y.backward(dy)
        res.append([y.detach().clone(), a.grad.clone(), b.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
    for u, v in zip(*res):
        assert torch.equal(u.contiguous(), v.contiguous())


def test_first_layer_takes_the_thin_input_kernels_emulated(emu, monkeypatch):
    """encoder1's UnetResBlock(4 -> 48) under bf16: conv1 through the thin-input kernels (forward + weight gradient) against
    the same block with the 48-channel kernels (SEGM_THIN_CONV_HIP=0 route): output and every parameter gradient agree"""
    from segmamba_amd import lib as L, unet_blocks as UB, conv3d as C3, fused_norm as FN
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    torch.manual_seed(5)
    blk = UB.UnetResBlock(4, 48)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, 2, 8, 64, generator=g).bfloat16()
    dy = torch.randn(1, 48, 2, 8, 64, generator=g).bfloat16()
    calls = []
    real = ops_raw.stem_conv_fwd
    monkeypatch.setattr(ops_raw, "stem_conv_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = []
    for thin in (True, False):
        monkeypatch.setattr(FN, "_THIN_HIP", thin)
        blk.zero_grad()
        y = blk(x)
        y.backward(dy)
        res.append([y.detach().float()] + [p.grad.clone() for p in blk.parameters()])
    assert len(calls) == 1                                # only the thin route calls it
    for u, v in zip(*res):
        assert u.dtype == v.dtype and (u - v).abs().max() <= 2e-2 * max(1.0, float(v.abs().max()))


def test_volume_empty_pads_power_of_two_channel_strides():
    """ops_raw.volume_empty: a 128^3 16-bit volume with >= 16 channels gets a padded channel stride (4 MiB strides alias in L2 /
    memory channels), everything else is an ordinary contiguous tensor; channel_dense recognises both"""
    v = ops_raw.volume_empty(1, 16, (128, 128, 128), torch.bfloat16, "cpu")
    assert v.shape == (1, 16, 128, 128, 128) and v.stride(1) == 128 ** 3 + 192 and v.stride(0) == 16 * v.stride(1)
    assert v.stride()[2:] == (128 * 128, 128, 1) and ops_raw.channel_dense(v) and not v.is_contiguous()
    assert v.flatten(2).data_ptr() == v.data_ptr()                                   # flattening the voxels stays a view
    for args in ((1, 4, (128, 128, 128), torch.bfloat16), (1, 48, (64, 64, 64), torch.bfloat16), (2, 48, (96, 96, 96), torch.float16)):
        t = ops_raw.volume_empty(*args, "cpu")
        assert t.is_contiguous() and ops_raw.channel_dense(t)
    assert ops_raw.volume_empty(1, 16, (128, 128, 128), torch.float16, "cpu").stride(1) == 128 ** 3 + 192
    assert ops_raw.volume_empty(1, 16, (64, 64, 64), torch.float32, "cpu").is_contiguous()               # 16-bit dtypes only
    one = torch.zeros(2, 1, 6 * 8 + 5)[:, :, :48].view(2, 1, 6, 8)                    # one channel: the batch stride is the instance stride
    assert ops_raw.channel_dense(one) and ops_raw.instance_stride(one) == 53 and ops_raw.instance_stride(v) == v.stride(1)
    assert not ops_raw.channel_dense(torch.zeros(2, 4, 6, 8).permute(0, 2, 1, 3))
    assert not ops_raw.channel_dense(torch.zeros(2, 8, 6, 8)[:, :3])                 # batch stride != channels * channel stride


@pytest.mark.parametrize("shape,dtype,with_add", [((2, 48, 200), torch.bfloat16, True), ((1, 130, 72), torch.float32, False),
                                                  ((1, 7, 13), torch.float32, True), ((2, 64, 64), torch.float16, False)])
def test_transpose_add_emulated(emu, shape, dtype, with_add):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dtype)
    add = torch.randn(shape[0], shape[2], shape[1], generator=g).to(dtype) if with_add else None
    out = ops_raw.transpose_add(emu, x, add)
    ref = x.transpose(1, 2).float() + (add.float() if with_add else 0)
    assert out.shape == (shape[0], shape[2], shape[1]) and out.is_contiguous()
    assert torch.equal(out, ref.to(dtype))


@pytest.mark.parametrize("dim,seqlen,chunk,channel_last,order,ns,dtype", [
    (32, 128, 32, True, L.TIME_FORWARD, 1, torch.float32),         # RW = 32, two work items per wave
    (64, 64, 32, True, L.TIME_REVERSED, 1, torch.float32),         # RW = 64
    (16, 256, 64, True, L.TIME_INTERLEAVED, 8, torch.float32),     # RW = 16, sub-tiles hop between slices
    (32, 128, 64, False, L.TIME_INTERLEAVED, 16, torch.float32),   # reference layout: time-fastest B / C staging
    (96, 64, 32, True, L.TIME_FORWARD, 1, torch.bfloat16),         # SegMamba stage-0 width
])
def test_scan_regular_shape_kernels_emulated(emu, monkeypatch, dim, seqlen, chunk, channel_last, order, ns, dtype):
    """scan_fwd_fast.hip / scan_bwd_fast.hip / scan_bwd_w8.hip (uniform addressing; the backward main kernel on 8-step windows,
    one workgroup per chunk group with a wave per d-tile) against the oracle AND against the general kernels."""
    c = H.scan_case(1, dim, 16, seqlen, dtype=dtype, seed=dim + seqlen)
    ref = H.scan_oracle(c, order, ns)
    monkeypatch.delenv("SEGM_SCAN_FAST", raising=False)
    fast = H.run_scan(emu, c, "cpu", channel_last, order, ns, chunk=chunk)
    H.check_scan(fast, ref, dtype, f"emu fast D={dim} L={seqlen}")
    monkeypatch.setenv("SEGM_SCAN_FAST", "0")
    slow = H.run_scan(emu, c, "cpu", channel_last, order, ns, chunk=chunk)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for k in ("out", "out_z"):
        if fast.get(k) is not None:
            assert (fast[k].float() - slow[k].float()).abs().max() <= tol * max(1.0, float(slow[k].float().abs().max())), k


@pytest.mark.parametrize("dim,seqlen,chunk,dtype", [(96, 128, 32, torch.bfloat16), (64, 128, 64, torch.float32), (20, 70, 32, torch.float32)])
def test_scan_three_directions_in_one_launch_emulated(emu, dim, seqlen, chunk, dtype):
    """segm_selective_scan_fwd_multi / _bwd_multi: the three directions of a Mamba v3 layer (forward, reversed, slice-interleaved;
    own parameters, inputs and outputs each) as ONE grid with a direction axis must equal three separate launches bit for bit
    (regular shapes), and fall back to separate launches for a ragged shape (dim 20, L 70)."""
    orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 8 if seqlen % 8 == 0 and chunk % 8 == 0 else 2)]
    cases = [H.to_dev_layout(H.scan_case(2, dim, 16, seqlen, dtype=dtype, seed=7 + i), "cpu", True) for i in range(3)]
    fcalls, single_f = [], []
    for d, (order, ns) in zip(cases, orders):
        kw = dict(u=d["u"], delta=d["delta"], A=d["A"], B=d["B"], C=d["C"], D=d["D"], z=d["z"], delta_bias=d["delta_bias"],
                  delta_softplus=True, channel_last=True, time_order=order, nslices=ns, chunk=chunk, need_out=True, need_ckpt=True)
        fcalls.append(kw)
        single_f.append(ops_raw.scan_fwd(emu, **kw))
    multi_f = ops_raw.scan_fwd_multi(emu, fcalls)
    for a, b in zip(single_f, multi_f):
        for k in ("out", "out_z", "ckpt"):
            assert torch.equal(a[k], b[k]), k
    bcalls, single_b = [], []
    for d, (order, ns), f in zip(cases, orders, single_f):
        kw = dict(u=d["u"], delta=d["delta"], A=d["A"], B=d["B"], C=d["C"], D=d["D"], z=d["z"], delta_bias=d["delta_bias"],
                  dout=d["g"], out=f["out"], ckpt=f["ckpt"], delta_softplus=True, channel_last=True, time_order=order,
                  nslices=ns, chunk=f["chunk"])
        bcalls.append(kw)
        single_b.append(ops_raw.scan_bwd(emu, **kw))
    multi_b = ops_raw.scan_bwd_multi(emu, bcalls)
    for a, b in zip(single_b, multi_b):
        for k in ("du", "ddelta", "dz", "dA", "dD", "ddelta_bias"):
            assert torch.equal(a[k], b[k]), k
        for k in ("dB", "dC"):
            if dim % 16 == 0:                                    # regular shapes: d-tiles summed in a fixed order inside one workgroup
                assert torch.equal(a[k], b[k]), k
            else:                                                # general kernels: several channel tiles add atomically
                assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-5), k


@pytest.mark.parametrize("with_bias", [True, False])
def test_conv1d_three_directions_in_one_launch_emulated(emu, with_bias):
    """segm_causal_conv1d_fwd_multi / _bwd_multi: the three directions of a Mamba v3 layer (own weights, one shared input) as one grid,
    their dW / db partials summed by ONE reduce launch with a direction axis (conv1d.hip reduce_partials_multi_kernel): outputs, dx, dW
    and db equal three separate calls bit for bit"""
    g = torch.Generator().manual_seed(41)
    B, Ln, D, W = 2, 256, 96, 4
    x = torch.randn(B, Ln, D, generator=g).bfloat16()
    orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 8)]
    ws = [torch.randn(D, W, generator=g) for _ in orders]
    bs = [torch.randn(D, generator=g) if with_bias else None for _ in orders]
    douts = [torch.randn(B, Ln, D, generator=g).bfloat16() for _ in orders]
    fcalls = [dict(x=x, weight=w, bias=b, silu=True, channel_last=True, time_order=o, nslices=ns) for w, b, (o, ns) in zip(ws, bs, orders)]
    single = [ops_raw.conv1d_fwd(emu, **c) for c in fcalls]
    multi = ops_raw.conv1d_fwd_multi(emu, fcalls)
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    bcalls = [dict(c, dout=d) for c, d in zip(fcalls, douts)]
    single_b = [ops_raw.conv1d_bwd(emu, c["x"], c["weight"], c["bias"], c["dout"], True, channel_last=True, time_order=c["time_order"],
                                   nslices=c["nslices"]) for c in bcalls]
    multi_b = ops_raw.conv1d_bwd_multi(emu, bcalls)
    for (dx0, dw0, db0), (dx1, dw1, db1) in zip(single_b, multi_b):
        assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1)
        assert (db0 is None and db1 is None) or torch.equal(db0, db1)


def test_scan_multi_call_with_different_flags_per_block_emulated(emu):
    """One segm_selective_scan_{fwd,bwd}_multi call whose blocks share the geometry but NOT the per-block flags - one without a
    gate, one without softplus, one with both, different time orders: the launch falls back to the kernels that read the flags
    per step (scan_fwd_fast.hip: fast_mode; scan_bwd_w8.hip: MODE 0) and must equal three separate launches bit for bit."""
    dim, seqlen, chunk = 64, 128, 32
    orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 8)]
    flags = [(False, True), (True, False), (True, True)]             # (gate, softplus)
    cases = [H.to_dev_layout(H.scan_case(2, dim, 16, seqlen, dtype=torch.float32, seed=21 + i), "cpu", True) for i in range(3)]
    fcalls, single_f = [], []
    for d, (order, ns), (gate, sp) in zip(cases, orders, flags):
        kw = dict(u=d["u"], delta=d["delta"], A=d["A"], B=d["B"], C=d["C"], D=d["D"], z=d["z"] if gate else None,
                  delta_bias=d["delta_bias"], delta_softplus=sp, channel_last=True, time_order=order, nslices=ns, chunk=chunk,
                  need_out=True, need_ckpt=True)
        fcalls.append(kw)
        single_f.append(ops_raw.scan_fwd(emu, **kw))
    multi_f = ops_raw.scan_fwd_multi(emu, fcalls)
    for a, b, (gate, _) in zip(single_f, multi_f, flags):
        for k in ("out", "ckpt") + (("out_z",) if gate else ()):
            assert torch.equal(a[k], b[k]), k
    bcalls, single_b = [], []
    for d, (order, ns), (gate, sp), f in zip(cases, orders, flags, single_f):
        kw = dict(u=d["u"], delta=d["delta"], A=d["A"], B=d["B"], C=d["C"], D=d["D"], z=d["z"] if gate else None,
                  delta_bias=d["delta_bias"], dout=d["g"], out=f["out"], ckpt=f["ckpt"], delta_softplus=sp, channel_last=True,
                  time_order=order, nslices=ns, chunk=f["chunk"])
        bcalls.append(kw)
        single_b.append(ops_raw.scan_bwd(emu, **kw))
    multi_b = ops_raw.scan_bwd_multi(emu, bcalls)
    for a, b, (gate, _) in zip(single_b, multi_b, flags):
        for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias") + (("dz",) if gate else ()):
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("dim,seqlen,chunk,order,ns,dtype", [
    (192, 64, 32, L.TIME_FORWARD, 1, torch.bfloat16),            # RW 64, three d-tiles (SegMamba stage 1)
    (384, 32, 16, L.TIME_REVERSED, 1, torch.float16),            # six d-tiles (stage 2)
    (768, 32, 16, L.TIME_INTERLEAVED, 8, torch.bfloat16),        # twelve (stage 3, BASELINE config 1)
    (48, 64, 16, L.TIME_FORWARD, 1, torch.float32),              # RW 16: four items per wave, three d-tiles
    (128, 48, 16, L.TIME_REVERSED, 1, torch.float32),            # RW 64, two d-tiles, chunk groups of one
])
def test_scan_backward_eight_step_windows_emulated(emu, dim, seqlen, chunk, order, ns, dtype):
    """scan_bwd_w8.hip over 1 - 12 d-tiles and the three lane groupings: every gradient against the oracle; dB / dC are sums over
    the d-tiles in a fixed order - bit-identical run to run - and, offered destinations of the tensors' own 16-bit type that are
    column windows of a wider matrix (the x_proj gradient operand), are written there once, rounded from the fp32 sum."""
    c = H.scan_case(2, dim, 16, seqlen, dtype=dtype, seed=dim)
    ref = H.scan_oracle(c, order, ns)
    res = H.run_scan(emu, c, "cpu", True, order, ns, chunk=chunk)
    H.check_scan(res, ref, dtype, f"emu w8 D={dim} L={seqlen}")
    d = H.to_dev_layout(c, "cpu", True)
    f = ops_raw.scan_fwd(emu, d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], True,
                         channel_last=True, time_order=order, nslices=ns, chunk=chunk, need_out=True, need_ckpt=True)
    kw = dict(channel_last=True, time_order=order, nslices=ns, chunk=f["chunk"])
    args = (d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["g"], f["out"], f["ckpt"], True)
    r1 = ops_raw.scan_bwd(emu, *args, **kw)
    r2 = ops_raw.scan_bwd(emu, *args, **kw)
    assert not r1["dbc_native"] and r1["dB"].dtype == torch.float32
    for k in ("dB", "dC", "du", "ddelta", "dA"):
        assert torch.equal(r1[k], r2[k]), k
    if dtype != torch.float32:
        wide = torch.full((2, seqlen, 40), 7.0, dtype=dtype)     # dt | pad | dB | dC | pad, as _rows_cols lays it out for R = 3
        r3 = ops_raw.scan_bwd(emu, *args, dB=wide[:, :, 4:20], dC=wide[:, :, 20:36], **kw)
        assert r3["dbc_native"]
        assert torch.equal(wide[:, :, 4:20], r1["dB"].to(dtype)) and torch.equal(wide[:, :, 20:36], r1["dC"].to(dtype))
        assert bool((wide[:, :, :4] == 7).all()) and bool((wide[:, :, 36:] == 7).all())      # nothing else is touched
        assert torch.equal(r3["du"], r1["du"]) and torch.equal(r3["dA"], r1["dA"])


def test_scan_backward_native_destinations_fall_back_on_irregular_shapes(emu):
    """a ragged shape takes the general kernels (atomic fp32 accumulation): 16-bit destinations are NOT written, the result
    says so and carries fp32 tensors; the C entry refuses `dbc_native` there instead of producing partial sums"""
    c = H.scan_case(1, 20, 16, 70, dtype=torch.bfloat16, seed=3)
    d = H.to_dev_layout(c, "cpu", True)
    f = ops_raw.scan_fwd(emu, d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], True,
                         channel_last=True, chunk=32, need_out=True, need_ckpt=True)
    dBn, dCn = torch.full((1, 70, 16), 7.0, dtype=torch.bfloat16), torch.full((1, 70, 16), 7.0, dtype=torch.bfloat16)
    r = ops_raw.scan_bwd(emu, d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["g"], f["out"],
                         f["ckpt"], True, channel_last=True, chunk=f["chunk"], dB=dBn, dC=dCn)
    assert not r["dbc_native"] and r["dB"].dtype == torch.float32 and bool((dBn == 7).all())
    a = L.ScanBwdArgs()
    ops_raw._scan_bwd_prepare(emu, a, d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["g"], f["out"],
                              f["ckpt"], True, channel_last=True, chunk=f["chunk"])
    assert emu.dll.segm_selective_scan_bwd_deterministic(a) == 0
    a.dbc_native = 1
    assert emu.dll.segm_selective_scan_bwd(a) == -2          # SEGM_E_SHAPE


def test_scan_rejects_views_beyond_32bit_offsets(emu):
    """L > 2^24 or a row stride >= 2^24 bytes is outside the kernels' 32-bit offset arithmetic: SEGM_E_SHAPE, nothing launched."""
    a = L.ScanFwdArgs()
    buf = torch.zeros(64)
    a.batch, a.dim, a.dstate, a.n_groups, a.seqlen = 1, 4, 16, 1, (1 << 24) + 16
    a.dtype, a.time_order, a.nslices, a.chunk = L.SEGM_F32, L.TIME_FORWARD, 1, 16
    for name in ("u", "delta", "out"):
        v = getattr(a, name)
        v.ptr, v.stride_b, v.stride_t, v.stride_d = buf.data_ptr(), 0, 4, 1
    for name in ("B", "C"):
        v = getattr(a, name)
        v.ptr, v.stride_b, v.stride_g, v.stride_t, v.stride_n = buf.data_ptr(), 0, 0, 16, 1
    a.A = buf.data_ptr()
    a.workspace, a.workspace_bytes = buf.data_ptr(), 1 << 40
    assert emu.dll.segm_selective_scan_fwd(a) == -2
    a.seqlen = 64
    a.u.stride_t = 1 << 23                                 # 2^23 elements * 4 bytes = 2^25 bytes per row
    assert emu.dll.segm_selective_scan_fwd(a) == -2


@pytest.mark.parametrize("shape", [(1, 48, 2, 3, 16), (1, 32, 3, 5, 64), (2, 16, 2, 4, 72), (1, 48, 1, 20, 8), (1, 96, 2, 3, 72)])
def test_conv3d_k3_fwd_emulated(emu, shape):
    """forward 3x3x3 convolution: transposed LDS staging, stationary weight fragments, kz reduction, zero padding."""
    B, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, 48, D, H_, W, generator=g).bfloat16()
    w = (0.1 * torch.randn(cout, 48, 3, 3, 3, generator=g)).bfloat16()
    bias = torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    y = ops_raw.conv3d_k3_fwd(emu, x, ops_raw.pack_conv3d_weight(w), bias)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    assert (y.float() - ref).abs().max() <= 1e-2 * max(1.0, float(ref.abs().max()))
    y0 = ops_raw.conv3d_k3_fwd(emu, x, ops_raw.pack_conv3d_weight(w))
    assert (y0.float() - (ref - bias.view(1, -1, 1, 1, 1))).abs().max() <= 1e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(1, 48, 2, 2, 16), (1, 48, 1, 16, 8), (1, 96, 1, 3, 72), (2, 48, 1, 2, 8)])
def test_conv3d_k3_fwd_chained_k_parts_emulated(emu, shape):
    """the pipelined variant (SEGM_CONV_FWD_CHAIN): K parts skewed in time, per-plane ring offsets, double-buffered
    hand-off, drain steps, y split; on the first shape also in-place accumulation of a second 48-channel input block
    (SEGM_CONV_FWD_ACCUMULATE) with both 48-channel kernels."""
    B, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, 96, D, H_, W, generator=g).bfloat16()
    w = (0.1 * torch.randn(cout, 96, 3, 3, 3, generator=g)).bfloat16()
    bias = torch.randn(cout, generator=g)
    ref0 = torch.nn.functional.conv3d(x[:, :48].float(), w[:, :48].float(), bias, 1, 1)
    tol = 1e-2 * max(1.0, float(ref0.abs().max()))
    y = ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:, :48]), bias, chain=True)
    assert (y.float() - ref0).abs().max() <= tol
    y48 = ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:, :48]), bias, chain=True, pitch48=True)
    assert torch.equal(y48, y)                                                                               # same sums, other LDS layout
    y32 = ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:, :48]), bias, chain32=True)
    assert torch.equal(y32, y)                                                                               # same sums, 32-wide x blocks
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    w1 = ops_raw.pack_conv3d_weight(w[:, 48:])
    # in-place accumulation of the second 48-channel input block with every chained kernel, on every shape (widths 8 / 16: x tiles
    # beyond W, whose lanes must neither read nor write - the old outputs are fetched at the head of the step since round 4)
    for kw in (dict(chain=True, pitch48=True), dict(chain32=True)):
        ya = y.clone()
        ops_raw.conv3d_k3_fwd(emu, x[:, 48:], w1, None, out=ya, accumulate=True, **kw)
        assert (ya.float() - ref).abs().max() <= 2 * tol, kw
    if shape != (1, 48, 2, 2, 16):
        return
    y2 = ops_raw.conv3d_k3_fwd(emu, x[:, 48:], w1, None, out=y, accumulate=True, chain=True)
    assert y2 is y and (y.float() - ref).abs().max() <= 2 * tol
    y3 = ref0.bfloat16()
    ops_raw.conv3d_k3_fwd(emu, x[:, 48:], w1, None, out=y3, accumulate=True)                 # the reduce-per-row kernel
    assert (y3.float() - ref).abs().max() <= 2 * tol
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:32, :48]), None, chain=True)      # Cout % 48 != 0
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_k3_fwd(emu, x[:, :48], ops_raw.pack_conv3d_weight(w[:, :48]), None, pitch48=True)      # needs chain


def test_conv3d_k3_dgrad_as_forward_emulated(emu):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 48, 2, 4, 16, generator=g, requires_grad=True)
    w = (0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
    dy = torch.randn(1, 48, 2, 4, 16, generator=g).bfloat16()
    torch.nn.functional.conv3d(x, w.float(), None, 1, 1).backward(dy.float())
    dx = ops_raw.conv3d_k3_fwd(emu, dy, ops_raw.pack_conv3d_weight_for_dgrad(w))
    assert (dx.float() - x.grad).abs().max() <= 1e-2 * max(1.0, float(x.grad.abs().max()))


@pytest.mark.parametrize("shape,dtype", [((2, 48, 200), torch.float32), ((1, 96, 72), torch.bfloat16), ((1, 192, 136), torch.float32),
                                         ((1, 384, 64), torch.bfloat16)])
def test_layernorm_tokens_emulated(emu, shape, dtype):
    B, Cc, S = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (1.5 * torch.randn(shape, generator=g) + 0.5).to(dtype)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    dy = torch.randn(B, S, Cc, generator=g).to(dtype)
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr.transpose(1, 2), (Cc,), gr, br, 1e-5)
    gx, gg, gb = torch.autograd.grad(ref, (xr, gr, br), dy.double())
    y, mean, rstd = ops_raw.layernorm_tokens_fwd(emu, x, gamma, beta, 1e-5)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert (y.double() - ref.detach()).abs().max() <= tol * max(1.0, float(ref.detach().abs().max()))
    dx, dgm, dbt = ops_raw.layernorm_tokens_bwd(emu, x, dy, mean, rstd, gamma)
    assert (dx.double() - gx).abs().max() <= tol * max(1.0, float(gx.abs().max()))
    assert (dgm.double() - gg).abs().max() <= 1e-3 * max(1.0, float(gg.abs().max()))
    assert (dbt.double() - gb).abs().max() <= 1e-3 * max(1.0, float(gb.abs().max()))


def test_conv3d_kernels_narrow_first_layer_emulated(emu):
    """4 input channels (SegMamba's first convolution): the kernels zero-fill the missing channels of the 48-channel block."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 4, 2, 5, 16, generator=g).bfloat16()
    w = (0.2 * torch.randn(48, 4, 3, 3, 3, generator=g)).bfloat16()
    dy = torch.randn(1, 48, 2, 5, 16, generator=g).bfloat16()
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    y = ops_raw.conv3d_k3_fwd(emu, x, ops_raw.pack_conv3d_weight(w))
    assert (y.float() - ref).abs().max() <= 1e-2 * max(1.0, float(ref.abs().max()))
    dw = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.float32)
    assert dw.shape == (48, 4, 3, 3, 3)
    ref_dw = _wgrad_reference(x, dy)
    assert (dw - ref_dw).abs().max() <= 1e-5 * ref_dw.abs().max() + 1e-4


def test_conv3d_kernels_fp16_emulated(emu):
    """the fp16 instantiations (v_mfma_f32_16x16x32_f16) of the forward and weight-gradient kernels"""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 48, 2, 4, 16, generator=g).half()
    w = (0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)).half()
    dy = torch.randn(1, 48, 2, 4, 16, generator=g).half()
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    y = ops_raw.conv3d_k3_fwd(emu, x, ops_raw.pack_conv3d_weight(w, torch.float16))
    assert y.dtype == torch.float16 and (y.float() - ref).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max()))
    dw = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.float16)
    ref_dw = _wgrad_reference(x, dy)
    assert dw.dtype == torch.float16 and (dw.float() - ref_dw).abs().max() <= 2e-3 * max(1.0, float(ref_dw.abs().max()))


def test_randomised_scan_sweep_emulated(emu):
    """a fixed-seed slice of tools/emu_random_sweep.py: random shapes / layouts / time orders / dtypes / optional arguments of
    the scan (forward and all gradients) against the oracle; half of the cases drawn from the regular-shape family."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["6", "3"], ["4", "5", "regular"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "emu_random_sweep.py"), *extra], capture_output=True,
                           text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]


def _torch_clip_sgd_reference(params, grads, steps, lr, mu, wd, nesterov, max_norm):
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.SGD(ps, lr=lr, momentum=mu, weight_decay=wd, nesterov=nesterov)
    norms = []
    for s in range(steps):
        for p, g in zip(ps, grads[s]):
            p.grad = g.clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(ps, max_norm)) if max_norm > 0 else None)
        opt.step()
    return [p.detach() for p in ps], norms


@pytest.mark.parametrize("max_norm,nesterov", [(0.5, True), (0.0, False)])
def test_sgd_clip_step_emulated(emu, max_norm, nesterov):
    """two-pass clip + SGD over a tensor list == clip_grad_norm_ + torch.optim.SGD: several launches' worth of tensors
    (> 96), sizes around the 16384-element workgroup chunk, an unaligned view, an empty tensor; two steps (momentum)."""
    g = torch.Generator().manual_seed(3)
    sizes = [1, 5, 16384, 16385, 40000, 0, 7] + [3 + i for i in range(100)]
    flat = torch.randn(sum(sizes) + 1, generator=g)
    params, off = [], 1                                    # views at odd offsets: not 16-byte aligned
    for n in sizes:
        params.append(flat[off:off + n].clone() if n % 2 else flat[off:off + n])
        off += n
    params = [p.contiguous() for p in params]
    grads = [[torch.randn(n, generator=g) for n in sizes] for _ in range(2)]
    lr, mu, wd = 0.05, 0.9, 1e-2
    want, norms = _torch_clip_sgd_reference(params, grads, 2, lr, mu, wd, nesterov, max_norm)
    mine = [p.clone() for p in params]
    mom = [torch.zeros_like(p) for p in mine]
    for s in range(2):
        head = ops_raw.sgd_clip_step(emu, mine, grads[s], mom, lr, mu, wd, nesterov, max_norm)
        if max_norm > 0:
            assert abs(float(head[1]) - norms[s]) <= 1e-5 * norms[s]
            assert abs(float(head[0]) - min(1.0, max_norm / (norms[s] + 1e-6))) <= 1e-5
    for a, b in zip(mine, want):
        assert a.shape == b.shape and (a.numel() == 0 or (a - b).abs().max() <= 1e-5 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize("shape,dtype", [((2, 4, 3, 5, 7), torch.float32), ((1, 13, 300), torch.bfloat16), ((3, 2, 10), torch.float16)])
def test_cross_entropy_emulated(emu, shape, dtype):
    g = torch.Generator().manual_seed(sum(shape))
    logits = (3 * torch.randn(shape, generator=g)).to(dtype)
    labels = torch.randint(0, shape[1], (shape[0],) + shape[2:], generator=g)
    labels.view(-1)[::7] = -100                           # ignored voxels
    ref_in = logits.double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, labels, reduction="sum")
    ref.backward()
    loss_sum, count, dlogits = ops_raw.cross_entropy(emu, logits, labels)
    assert float(count) == float((labels != -100).sum())
    assert abs(float(loss_sum) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    tol = 1e-6 if dtype == torch.float32 else (1e-3 if dtype == torch.float16 else 8e-3)
    assert dlogits.dtype == dtype and (dlogits.double() - ref_in.grad).abs().max() <= tol
    # a label outside [0, C) that is not ignore_index must not pass silently (ATen asserts): the loss becomes NaN
    bad = labels.clone()
    bad.view(-1)[1] = shape[1]
    loss_bad, _, d_bad = ops_raw.cross_entropy(emu, logits, bad)
    assert torch.isnan(loss_bad) and torch.isnan(d_bad.float()).any()


def test_conv_dispatcher_library_routes_on_emulated_kernels(emu, monkeypatch):
    """conv3d.py's library candidates (forward / data gradient; reduce-per-row, chained, chained with unpadded LDS rows;
    96 input channels = two blocks accumulated in place) produce the convolution, through the emulated kernels."""
    from segmamba_amd import conv3d as C3, lib as L
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1, 96, 1, 2, 8, generator=g).bfloat16()
    w = (0.1 * torch.randn(48, 96, 3, 3, 3, generator=g)).bfloat16()
    bias = torch.randn(48, generator=g).bfloat16()
    dy = torch.randn(1, 48, 1, 2, 8, generator=g).bfloat16()
    xr = x.float().requires_grad_()
    ref = torch.nn.functional.conv3d(xr, w.float(), bias.float(), 1, 1)
    ref.backward(dy.float())
    assert C3._hip_fwd_ok(x, w) and C3._hip_chain_ok(w)
    for chain, p48, c32 in ((False, False, False), (True, False, False), (True, True, False), (False, False, True)):
        y = C3._fwd_hip(x, w, 1, bias, chain, p48, c32)
        assert (y.float() - ref.detach()).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
    w2 = w[:, :48].contiguous()                            # data gradient: 48 -> 48 (flipped weights: Cout' = 48)
    x2 = x[:, :48].float().requires_grad_()
    torch.nn.functional.conv3d(x2, w2.float(), None, 1, 1).backward(dy.float())
    for chain, p48, c32 in ((False, False, False), (True, True, False), (False, False, True)):
        dx = C3._dgrad_hip(dy, w2, x[:, :48], 1, chain, p48, c32)
        assert (dx.float() - x2.grad).abs().max() <= 2e-2 * max(1.0, float(x2.grad.abs().max()))


@pytest.mark.parametrize("dtype,state_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
def test_decode_step_kernels_emulated(emu, dtype, state_dtype):
    """segm_causal_conv1d_update / segm_selective_state_update against the golden fixture of the reference's *_ref functions
    (fp32) and against the oracle on strided, lower-precision inputs."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mamba_decode.npz"))
    f = {k: torch.from_numpy(z[k]) for k in z.files}
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    # conv update: x as a strided half of a wider tensor, as Mamba.step passes it
    xz = torch.cat([f["cu.x"], f["cu.x"] + 1], dim=1).to(dtype)
    x = xz[:, :f["cu.x"].shape[1]]
    cs = f["cu.state_in"].to(dtype).clone()
    cs_ref = cs.float().clone()
    want = ref_ops.causal_conv1d_update_ref(x.float(), cs_ref, f["cu.weight"], f["cu.bias"], "silu")
    got = ops_raw.conv1d_update(emu, x, cs, f["cu.weight"], f["cu.bias"], True)
    assert got.dtype == dtype and (got.float() - want).abs().max() <= tol * max(1.0, float(want.abs().max()))
    assert torch.equal(cs.float(), cs_ref.to(dtype).float())
    if dtype == torch.float32:
        assert (got - f["cu.out"]).abs().max() <= 2e-5 and (cs - f["cu.state_out"]).abs().max() == 0
    # state update
    st = f["su.state_in"].to(state_dtype).clone()
    st_ref = st.clone()
    args = [f[k].to(dtype) for k in ("su.x", "su.dt")]
    Bm, Cm, zz = f["su.B"].to(dtype), f["su.C"].to(dtype), f["su.z"].to(dtype)
    want = ref_ops.selective_state_update_ref(st_ref, args[0].float(), args[1].float(), f["su.A"], Bm.float(), Cm.float(), f["su.D"],
                                              z=zz.float(), dt_bias=f["su.dt_bias"], dt_softplus=True)
    got = ops_raw.state_update(emu, st, args[0], args[1], f["su.A"], Bm, Cm, f["su.D"], zz, f["su.dt_bias"], True)
    assert (got.float() - want).abs().max() <= tol * max(1.0, float(want.abs().max()))
    assert (st.float() - st_ref.float()).abs().max() <= (2e-5 if state_dtype == torch.float32 else 2e-2) * max(1.0, float(st_ref.abs().max()))
    if dtype == torch.float32:
        assert (got - f["su.out"]).abs().max() <= 2e-5 * max(1.0, float(f["su.out"].abs().max()))
        assert (st - f["su.state_out"]).abs().max() <= 2e-5 * max(1.0, float(f["su.state_out"].abs().max()))
    with pytest.raises(RuntimeError):
        ops_raw.conv1d_update(emu, x, torch.zeros(3, 10, 5, dtype=dtype), torch.zeros(10, 5), None, True)     # width 5


def test_mamba_decode_host_path_on_emulated_kernels(emu, monkeypatch):
    """`Mamba.forward(h, inference_params)` (prefill at seqlen_offset 0, then `step` per token) through the product's host
    code with the kernels emulated == the reference Mamba's own run (tests/golden/make_golden_decode.py)."""
    import types
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm import Mamba
    f = H.load_golden("mamba_decode.npz")
    m = Mamba(d_model=12, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=4, layer_idx=0)
    m.load_state_dict({k[len("param."):]: v for k, v in f.items() if k.startswith("param.")})
    h, L0 = f["h"], int(f["L0"])
    params = types.SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0)
    with torch.no_grad():
        out = m(h[:, :L0], inference_params=params)
        conv, ssm = params.key_value_memory_dict[0]
        H.assert_close(out, f["out_prefill"], 1e-4, 1e-5, "prefill out")
        H.assert_close(conv, f["conv_state_prefill"], 1e-5, 1e-6, "prefill conv_state")
        H.assert_close(ssm, f["ssm_state_prefill"], 1e-4, 1e-5, "prefill ssm_state")
        outs = []
        for t in range(L0, h.shape[1]):
            params.seqlen_offset = t
            outs.append(m(h[:, t:t + 1], inference_params=params))
    H.assert_close(torch.cat(outs, 1), f["out_steps"], 1e-4, 1e-5, "step outs")
    H.assert_close(conv, f["conv_state_final"], 1e-5, 1e-6, "final conv_state")
    H.assert_close(ssm, f["ssm_state_final"], 1e-4, 1e-5, "final ssm_state")
    c2, s2 = m.allocate_inference_cache(3, 0)
    assert c2.shape == (3, 24, 4) and s2.shape == (3, 24, 16) and c2.dtype == torch.float32


def test_randomised_conv3d_forward_sweep_emulated(emu):
    """fixed-seed random shapes through every forward-convolution kernel: narrow / full input blocks, 16 / 32 / 48 / 96 output
    channels, widths that end inside an x block, heights that force a y split, strided channel slices, all three 48-channel
    kernel variants and in-place accumulation."""
    rng = np.random.default_rng(1234)
    for case in range(10):
        B = int(rng.integers(1, 3))
        cin = int(rng.choice([3, 17, 48, 48, 48]))
        cout = int(rng.choice([16, 32, 48, 48, 96]))
        D, H_ = int(rng.integers(1, 4)), int(rng.choice([1, 2, 5, 17]))
        W = int(rng.choice([8, 16, 24, 72]))
        g = torch.Generator().manual_seed(100 + case)
        full = torch.randn(B, cin + 5, D, H_, W, generator=g).bfloat16()
        x = full[:, 2:2 + cin]                                             # a strided channel slice
        w = (0.2 * torch.randn(cout, cin, 3, 3, 3, generator=g)).bfloat16()
        bias = torch.randn(cout, generator=g)
        ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
        tol = 1e-2 * max(1.0, float(ref.abs().max()))
        wp = ops_raw.pack_conv3d_weight(w)
        variants = [dict()] + ([dict(chain=True), dict(chain=True, pitch48=True), dict(chain32=True)] if cout % 48 == 0 else [])
        for kw in variants:
            y = ops_raw.conv3d_k3_fwd(emu, x, wp, bias, **kw)
            assert (y.float() - ref).abs().max() <= tol, (case, kw, tuple(x.shape), cout)
            if cout % 48 == 0:
                y2 = ops_raw.conv3d_k3_fwd(emu, x, wp, None, out=y.clone(), accumulate=True, **kw)
                want = y.float() + (ref - bias.view(1, -1, 1, 1, 1))
                assert (y2.float() - want).abs().max() <= 2 * tol, (case, kw, "accumulate")


@pytest.mark.parametrize("dstate,groups", [(40, 1), (32, 2)])
def test_selective_scan_fn_wide_state_on_emulated_kernels(emu, monkeypatch, dstate, groups):
    """dstate > 16 (the reference takes up to 256): blocks of 16 states summed by the host wrapper, against the oracle's
    selective_scan_ref - output, last state and every gradient."""
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    g = torch.Generator().manual_seed(dstate)
    Bsz, D, Lq = 1, 4, 24
    shape_bc = (Bsz, groups, dstate, Lq) if groups > 1 else (Bsz, dstate, Lq)
    t = {"u": torch.randn(Bsz, D, Lq, generator=g), "delta": 0.5 * torch.rand(Bsz, D, Lq, generator=g),
         "A": -0.5 * torch.rand(D, dstate, generator=g) - 0.05, "B": torch.randn(shape_bc, generator=g),
         "C": torch.randn(shape_bc, generator=g), "D": torch.randn(D, generator=g), "z": torch.randn(Bsz, D, Lq, generator=g),
         "delta_bias": 0.5 * torch.rand(D, generator=g)}
    dout = torch.randn(Bsz, D, Lq, generator=g)
    res = {}
    for name, fn in (("hip", selective_scan_fn), ("ref", ref_ops.selective_scan_ref)):
        leaves = {k: v.clone().requires_grad_() for k, v in t.items()}
        out, last = fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], z=leaves["z"],
                       delta_bias=leaves["delta_bias"], delta_softplus=True, return_last_state=True)
        out.backward(dout)
        res[name] = (out.detach(), last.detach(), {k: v.grad for k, v in leaves.items()})
    H.assert_close(res["hip"][0], res["ref"][0], 1e-4, 1e-4, "out")
    H.assert_close(res["hip"][1], res["ref"][1], 1e-4, 1e-4, "last_state")
    for k in t:
        r = res["ref"][2][k]
        H.assert_close(res["hip"][2][k], r, 1e-3, 1e-3 * max(1.0, float(r.abs().max())), "d" + k)


@pytest.mark.parametrize("M,K,N,dtype,bias", [(100, 48, 192, torch.bfloat16, False), (37, 96, 48, torch.bfloat16, True),
                                              (200, 192, 100, torch.float16, True), (16, 128, 388, torch.bfloat16, False),
                                              (33, 40, 36, torch.bfloat16, True), (64, 48, 200, torch.bfloat16, True),
                                              (50, 96, 104, torch.float16, False),
                                              # round 6: K > 192 (streamed W): odd and even chunk counts, k tail, row / column tails
                                              (70, 384, 44, torch.bfloat16, False), (33, 768, 200, torch.bfloat16, True),
                                              (129, 200, 100, torch.float16, True), (16, 1536, 96, torch.bfloat16, False)])
def test_linear_rows_emulated(emu, M, K, N, dtype, bias):
    """row-streaming projection: every K-chunk count (K <= 64 / 96 / 128 / 192), ragged row and column tails, several column
    blocks, strided input rows (a column slice of a wider tensor) and output rows (written into a wider tensor); K > 192: the
    kernel that streams W as well."""
    g = torch.Generator().manual_seed(M + K + N)
    xw = torch.randn(M, K + 16, generator=g).to(dtype)
    x = xw[:, 8:8 + K]                                       # 16-byte aligned column slice
    w = (0.2 * torch.randn(N, K, generator=g)).to(dtype)
    b = torch.randn(N, generator=g) if bias else None
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    yw = torch.full((M, N + 8), 7.0).to(dtype)
    y = ops_raw.linear_rows(emu, x, w, b, out=yw[:, 4:4 + N])
    tol = (1e-2 if dtype == torch.bfloat16 else 2e-3) * max(1.0, float(ref.abs().max()))
    assert (y.float() - ref).abs().max() <= tol
    assert (yw[:, :4] == 7).all() and (yw[:, 4 + N:] == 7).all()          # nothing outside the slice is touched
    y2 = ops_raw.linear_rows(emu, x.contiguous(), w, b)       # N % 8 == 0: the 16-byte-store form (tile pairs); the slice above: 8-byte
    assert torch.equal(y2, yw[:, 4:4 + N])
    if K > 192:
        tol = tol * (K / 192.0) ** 0.5                        # the reference sum grows with sqrt(K); so does one rounding of it
    y3 = ops_raw.linear_rows(emu, x, w, None, out=y2.clone(), accumulate=True)                  # y3 = y2 + x W^T
    want = y2.float() + torch.nn.functional.linear(x.float(), w.float())
    assert (y3.float() - want).abs().max() <= 2 * tol
    assert not ops_raw.linear_rows_supported(x.float(), w.float()) and not ops_raw.linear_rows_supported(xw[:, 1:1 + K], w)


def test_linear_cl_library_route_on_emulated_kernels(emu, monkeypatch):
    """linear.linear_cl with the row-streaming kernel switched in (SEGM_LINEAR_HIP): forward and data gradient through the
    library, weight gradient through the split-K path - against F.linear autograd."""
    from segmamba_amd import linear as LN, lib as Lm, ops_raw as OR
    monkeypatch.setattr(Lm, "_lib", emu)
    monkeypatch.setattr(LN, "_ROWS_HIP", True)
    monkeypatch.setattr(LN, "_ROWS_MIN", 1)
    calls = []
    real = OR.linear_rows
    monkeypatch.setattr(OR, "linear_rows", lambda *a, **k: (calls.append(a[1].shape), real(*a, **k))[1])
    monkeypatch.setattr(LN, "_on_device", lambda t: True)                             # the route is taken for CUDA tensors only
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 20, 48, generator=g).bfloat16().requires_grad_()
    w = (0.2 * torch.randn(96, 48, generator=g)).bfloat16().requires_grad_()
    dy = torch.randn(2, 20, 96, generator=g).bfloat16()
    y = LN._LinearCL.apply(x, w, None)
    gx, gw = torch.autograd.grad(y, (x, w), dy)
    monkeypatch.undo()
    x2, w2 = x.detach().float().requires_grad_(), w.detach().float().requires_grad_()
    y2 = torch.nn.functional.linear(x2, w2)
    gx2, gw2 = torch.autograd.grad(y2, (x2, w2), dy.float())
    assert len(calls) == 2 and calls[0] == (40, 48) and calls[1] == (40, 96)         # forward and data gradient
    for got, want in ((y, y2), (gx, gx2), (gw, gw2)):
        assert (got.float() - want).abs().max() <= 2e-2 * max(1.0, float(want.abs().max()))


def test_mamba_block_with_library_projections_on_emulated_kernels(emu, monkeypatch):
    """SEGM_LINEAR_HIP route: in / out / x / dt projections and their data gradients through segm_linear_rows (padded x_dbl,
    accumulate into dconv) == the BLAS route, on a bf16 Mamba(v3) block: output, input gradient, all 23 parameter gradients."""
    from segmamba_amd import linear as LN
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm import Mamba
    from tests.golden.make_golden import named_fill
    m = Mamba(d_model=16, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=4)
    m.load_state_dict(named_fill(m.state_dict()))
    m = m.bfloat16()
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(2, 24, 16, generator=g).bfloat16()
    dy = torch.randn(2, 24, 16, generator=g).bfloat16()
    res, calls = [], []
    real = ops_raw.linear_rows
    monkeypatch.setattr(ops_raw, "linear_rows", lambda *a, **k: (calls.append((tuple(a[1].shape), tuple(a[2].shape))), real(*a, **k))[1])
    from segmamba_amd import selective_scan_interface as SSI
    for route, recompute in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(SSI, "_RECOMPUTE", recompute)
        monkeypatch.setattr(LN, "_ROWS_HIP", route)
        monkeypatch.setattr(LN, "_ROWS_MIN", 1)
        monkeypatch.setattr(LN, "_on_device", lambda t: True)
        assert route or not calls                          # nothing goes through the kernel while the route is off
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        y = m(x)
        y.backward(dy)
        res.append((y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()}))
    # per direction: x_proj, dt_proj (forward), ddelta @ W_dt, dx_dbl @ W_x - and dt_proj again when the backward recomputes
    # delta (SEGM_RECOMPUTE=1, the reference's checkpoint_lvl 1); plus in / out proj x 2
    assert len(calls) == (3 * 4 + 4) + (3 * 5 + 4), len(calls)
    (y0, gx0, gp0) = res[0]
    for y1, gx1, gp1 in res[1:]:
        assert (y1 - y0).abs().max() <= 3e-2 * max(1.0, float(y0.abs().max()))
        assert (gx1 - gx0).abs().max() <= 3e-2 * max(1.0, float(gx0.abs().max()))
        for k in gp0:
            assert (gp1[k] - gp0[k]).abs().max() <= 5e-2 * max(1e-2, float(gp0[k].abs().max())), k


def test_conv_same_autograd_with_every_library_candidate_on_emulated_kernels(emu, monkeypatch):
    """conv3d._ConvSame (forward, data gradient, weight gradient, bias gradient) with each of the dispatcher's library
    candidates forced in turn - the CPU twin of tests/test_gpu_kernels.py::test_conv3d_same_autograd_with_library_kernels."""
    from segmamba_amd import conv3d as C3
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setenv("SEGM_CONV_FWD_UNTIMED", "1")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 48, 2, 4, 16, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16().requires_grad_()
    bias = torch.randn(48, generator=g).bfloat16().requires_grad_()
    dy = torch.randn(1, 48, 2, 4, 16, generator=g).bfloat16()
    x2, w2, b2 = (t.detach().float().requires_grad_() for t in (x, w, bias))
    want = torch.autograd.grad(torch.nn.functional.conv3d(x2, w2, b2, 1, 1), (x2, w2, b2), dy.float())
    y_want = torch.nn.functional.conv3d(x2, w2, b2, 1, 1).detach()
    for idx in (-1, -2, -3, -4):                           # chain32, chained + unpadded rows, chained, reduce-per-row
        monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest, idx=idx: cands[max(idx, -len(cands))]())
        y = C3._ConvSame.apply(x, w, bias)
        got = torch.autograd.grad(y, (x, w, bias), dy)
        for a, b in zip((y,) + got, (y_want,) + want):
            assert (a.float() - b).abs().max() <= 2e-2 * max(1.0, float(b.abs().max())), idx


def test_segmamba_forward_golden_through_library_routes_on_emulated_kernels(emu, monkeypatch):
    """The whole network in fp32 with the library's routes switched on for CPU tensors (InstanceNorm, transpose + LayerNorm,
    transposes, conv1d, scans - the MFMA convolutions are 16-bit only and stay on ATen here) against the reference SegMamba's
    own forward (vendored MONAI blocks + reference Mamba on its ref ops): the CPU twin of the GPU golden test."""
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    from model_segmamba.segmamba import SegMamba
    from tests.golden.make_golden import named_fill
    f = H.load_golden("segmamba_tiny.npz")
    m = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 8, 16, 32], hidden_size=32)
    sd = named_fill(m.state_dict())
    assert len(sd) == int(f["nkeys"])
    m.load_state_dict(sd)
    m = m.eval()
    x = torch.rand(1, 4, 32, 32, 32, generator=torch.Generator().manual_seed(int(f["x_seed"])))
    with torch.no_grad():
        y = m(x)
    H.assert_close(y[:, :, ::2, ::2, ::2], f["y_sub"], 2e-3, 2e-3, "y_sub")
    assert abs(float(y.mean()) - float(f["y_mean"])) < 1e-3 and abs(float(y.std()) - float(f["y_std"])) < 1e-3


def test_segmamba_bf16_forward_with_library_convolutions_on_emulated_kernels(emu, monkeypatch):
    """As above in bf16 and with the dispatcher's last forward candidate (the chained MFMA kernel on 32-wide x blocks) forced
    for every 3x3x3 convolution it applies to: close to the reference's fp32 forward at bf16 precision."""
    from segmamba_amd import conv3d as C3
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setenv("SEGM_CONV_FWD_UNTIMED", "1")
    routed = []
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: (routed.append(len(cands)), cands[-1]())[1])
    from model_segmamba.segmamba import SegMamba
    from tests.golden.make_golden import named_fill
    f = H.load_golden("segmamba_tiny.npz")
    m = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 8, 16, 32], hidden_size=32)
    m.load_state_dict(named_fill(m.state_dict()))
    m = m.bfloat16().eval()
    x = torch.rand(1, 4, 32, 32, 32, generator=torch.Generator().manual_seed(int(f["x_seed"])))
    with torch.no_grad():
        y = m(x.bfloat16()).float()
    assert max(routed) >= 5                                # native, reduce-per-row, chained, chained unpadded, chained 32-wide
    H.assert_close(y[:, :, ::2, ::2, ::2], f["y_sub"], 6e-2, 6e-2, "y_sub (bf16)")


def test_randomised_norm_layout_wgrad_sweep_emulated(emu):
    """fixed-seed random shapes through the stem's other kernels: InstanceNorm (+ residual) (+ activation) forward and backward,
    transpose (+ add), transpose + LayerNorm forward and backward, and the MFMA weight gradient."""
    rng = np.random.default_rng(77)
    F = torch.nn.functional
    for case in range(24):
        g = torch.Generator().manual_seed(500 + case)
        dtype = [torch.float32, torch.bfloat16, torch.float16][case % 3]
        tol = {torch.float32: 2e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype]
        # ---- InstanceNorm ------------------------------------------------------------------------------------------
        B, Cc = int(rng.integers(1, 3)), int(rng.integers(1, 7))
        sp = tuple(int(v) for v in rng.integers(1, 9, size=3))
        if int(np.prod(sp)) < 2:
            sp = (2, 1, 1)
        act = ["none", "relu", "leaky_relu"][int(rng.integers(0, 3))]
        use_res = bool(rng.integers(0, 2))
        x = (2 * torch.randn(B, Cc, *sp, generator=g) + 0.5).to(dtype)
        res = torch.randn(B, Cc, *sp, generator=g).to(dtype) if use_res else None
        dy = torch.randn(B, Cc, *sp, generator=g).to(dtype)
        xr = x.double().requires_grad_()
        rr = res.double().requires_grad_() if use_res else None
        pre = F.instance_norm(xr, eps=1e-5) + (rr if use_res else 0)
        ref = {"none": lambda v: v, "relu": F.relu, "leaky_relu": lambda v: F.leaky_relu(v, 0.01)}[act](pre)
        grads = torch.autograd.grad(ref, (xr, rr) if use_res else (xr,), dy.double())
        ref = ref.detach()
        y, mean, rstd = ops_raw.instnorm_fwd(emu, x, res, act)
        assert (y.double() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max())), ("instnorm fwd", case)
        dx, dres = ops_raw.instnorm_bwd(emu, x, dy, mean, rstd, y if (act != "none" and use_res) else None, act, want_dresidual=use_res)
        # elements that sit exactly at an activation kink after rounding may differ: compare in the bulk
        bad = ((dx.double() - grads[0]).abs() > 4 * tol * max(1.0, float(grads[0].abs().max()))).float().mean()
        assert bad <= (0.0 if dtype == torch.float32 else 0.02), ("instnorm bwd dx", case, float(bad))
        if use_res:
            bad = ((dres.double() - grads[1]).abs() > 4 * tol * max(1.0, float(grads[1].abs().max()))).float().mean()
            assert bad <= (0.0 if dtype == torch.float32 else 0.02), ("instnorm bwd dres", case)
        # ---- transpose (+ add) -------------------------------------------------------------------------------------
        R, Ct = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        xt = torch.randn(B, R, Ct, generator=g).to(dtype)
        add = torch.randn(B, Ct, R, generator=g).to(dtype) if use_res else None
        want = xt.transpose(1, 2).float() + (add.float() if use_res else 0)
        got = ops_raw.transpose_add(emu, xt, add)
        assert (got.float() - want).abs().max() <= tol * max(1.0, float(want.abs().max())), ("transpose", case)
        # ---- transpose + LayerNorm ---------------------------------------------------------------------------------
        n = 4 if dtype == torch.float32 else 8
        Cl, S = n * int(rng.integers(1, 13)), n * int(rng.integers(1, 30))
        xl = (1.5 * torch.randn(B, Cl, S, generator=g) + 0.3).to(dtype)
        gamma, beta = torch.randn(Cl, generator=g), torch.randn(Cl, generator=g)
        dyl = torch.randn(B, S, Cl, generator=g).to(dtype)
        xlr, gr, br = xl.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
        refl = F.layer_norm(xlr.transpose(1, 2), (Cl,), gr, br, 1e-5)
        gx, gg, gb = torch.autograd.grad(refl, (xlr, gr, br), dyl.double())
        refl = refl.detach()
        yl, m2, r2 = ops_raw.layernorm_tokens_fwd(emu, xl, gamma, beta, 1e-5)
        assert (yl.double() - refl).abs().max() <= 2 * tol * max(1.0, float(refl.abs().max())), ("layernorm fwd", case)
        dxl, dg, db = ops_raw.layernorm_tokens_bwd(emu, xl, dyl, m2, r2, gamma)
        assert (dxl.double() - gx).abs().max() <= 2 * tol * max(1.0, float(gx.abs().max())), ("layernorm dx", case)
        assert (dg.double() - gg).abs().max() <= max(tol, 2e-3) * max(1.0, float(gg.abs().max())), ("layernorm dgamma", case)
        assert (db.double() - gb).abs().max() <= max(tol, 2e-3) * max(1.0, float(gb.abs().max())), ("layernorm dbeta", case)
    # ---- MFMA weight gradient ---------------------------------------------------------------------------------------
    for case in range(8):
        g = torch.Generator().manual_seed(900 + case)
        B = int(rng.integers(1, 3))
        cin = int(rng.choice([4, 30, 48, 96]))
        cout = int(rng.choice([48, 96]))
        D, H_, W = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.choice([8, 16, 40, 72]))
        dt = torch.bfloat16 if case % 2 == 0 else torch.float16
        xw = torch.randn(B, cin, D, H_, W, generator=g).to(dt)
        dyw = torch.randn(B, cout, D, H_, W, generator=g).to(dt)
        dw = ops_raw.conv3d_k3_wgrad(emu, xw, dyw, torch.float32)
        ref_dw = _wgrad_reference(xw, dyw)
        assert dw.shape == ref_dw.shape and (dw - ref_dw).abs().max() <= 1e-5 * ref_dw.abs().max() + 1e-3, ("wgrad", case, cin, cout, D, H_, W)


@pytest.mark.parametrize("B,Cin,Cout,S,dtype,bias,view", [(2, 48, 48, 128, torch.bfloat16, True, False), (1, 4, 48, 64, torch.bfloat16, False, False),
                                                        (2, 48, 4, 192, torch.float16, True, False), (1, 96, 48, 64, torch.bfloat16, True, True),
                                                        (1, 40, 96, 128, torch.bfloat16, False, False)])
def test_pointwise_cf_emulated(emu, B, Cin, Cout, S, dtype, bias, view):
    """segm_pointwise_cf (channel-first 1x1x1 convolution: LDS transpose of the voxel strips, stationary weights, bias in the
    accumulator) against fp32 matmul on the same 16-bit operands; narrow / ragged channel counts, a channel-slice view of a wider
    tensor as input, accumulation into an existing result"""
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    full = torch.randn(B, Cin + (16 if view else 0), S, generator=g).to(dtype)
    x = full[:, 8:8 + Cin] if view else full
    w = (0.2 * torch.randn(Cout, Cin, generator=g)).to(dtype)
    b = torch.randn(Cout, generator=g) if bias else None
    assert ops_raw.pointwise_cf_supported(x, Cout)
    y = ops_raw.pointwise_cf(emu, x, w, b)
    ref = torch.einsum("oc,bcs->bos", w.float(), x.float()) + (b.view(1, -1, 1) if bias else 0)
    assert y.shape == (B, Cout, S) and y.dtype == dtype
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    y2 = ops_raw.pointwise_cf(emu, x, w, None, out=y.clone(), accumulate=True)
    ref2 = y.float() + torch.einsum("oc,bcs->bos", w.float(), x.float())
    assert (y2.float() - ref2).abs().max() <= tol * max(1.0, float(ref2.abs().max()))
    # the result above went out in 16-byte stores (rows of the fresh output are 16-byte aligned: voxel blocks dealt in pairs); a
    # destination whose channel rows are only 8-byte aligned takes the 8-byte form: the same values, nothing outside written
    buf = torch.full((B, Cout, S + 4), 7.0).to(dtype)
    y8 = ops_raw.pointwise_cf(emu, x, w, b, out=buf[:, :, :S])
    assert torch.equal(y8, y) and (buf[:, :, S:] == 7).all()
    y8b = ops_raw.pointwise_cf(emu, x, w, None, out=y8, accumulate=True)
    assert torch.equal(y8b, y2)
    with pytest.raises(RuntimeError):
        ops_raw.pointwise_cf(emu, x[:, :, :S - 8], w)                      # voxels not a multiple of 64


def test_pointwise_autograd_route_on_emulated_kernels(emu, monkeypatch):
    """linear.pointwise (the 1x1x1 convolutions of the conv stem) with the library kernel switched in: forward with the bias
    fused and the data gradient through segm_pointwise_cf == the BLAS route, output and all three gradients"""
    from segmamba_amd import linear as LN
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(LN, "_on_device", lambda t: True)
    monkeypatch.setattr(LN, "_PW_MIN", 1)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(2, 48, 4, 4, 8, generator=g).bfloat16()
    w0 = (0.2 * torch.randn(96, 48, generator=g)).bfloat16()
    b0 = torch.randn(96, generator=g).bfloat16()
    dy = torch.randn(2, 96, 4, 4, 8, generator=g).bfloat16()
    res = []
    for on in (False, True):
        monkeypatch.setattr(LN, "_PW_HIP", on)
        x, w, b = x0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        y = LN.pointwise(x, w, b)
        y.backward(dy)
        res.append([t.float() for t in (y.detach(), x.grad, w.grad, b.grad)])
    for a, c in zip(*res):
        assert (a - c).abs().max() <= 3e-2 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,D,H,W,dtype,bias", [(1, 4, 48, 4, 16, 32, torch.bfloat16, True), (2, 1, 16, 2, 8, 64, torch.bfloat16, False),
                                                       (1, 3, 32, 6, 16, 32, torch.float16, True), (1, 4, 48, 2, 4, 128, torch.bfloat16, False),
                                                       (1, 2, 16, 2, 2, 256, torch.bfloat16, True)])
def test_stem_conv_fwd_emulated(emu, monkeypatch, B, Cin, Cout, D, H, W, dtype, bias):
    """segm_stem_conv_fwd (7^3, stride 2, padding 3: implicit GEMM with K = (kz, ky, kx slot, ci), channel-last-4 input, packed
    weights) against torch's conv3d in fp32 on the same 16-bit operands: every tile-block shape (TX x TY), fewer than 4 input
    channels, fewer than 48 output channels, volumes smaller than the kernel (all padding cases)"""
    g = torch.Generator().manual_seed(B + Cin + Cout + D)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dtype)
    w = (0.05 * torch.randn(Cout, Cin, 7, 7, 7, generator=g)).to(dtype)
    b = torch.randn(Cout, generator=g) if bias else None
    assert ops_raw.stem_conv_supported(x, w)
    y = ops_raw.stem_conv_fwd(emu, x, w, b)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), b, stride=2, padding=3)
    assert y.shape == ref.shape and y.dtype == dtype
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    # tiles that pair along x (output width >= 32) leave in 16-byte stores; the 8-byte form gives the same values
    monkeypatch.setenv("SEGM_STEM_WIDE", "0")
    assert torch.equal(ops_raw.stem_conv_fwd(emu, x, w, b), y)


def test_thin_input_conv3_emulated(emu):
    """the same two kernels as the 3x3x3 stride-1 padding-1 convolution on <= 4 input channels (UnetResBlock conv1 of encoder1):
    forward (with a padded output channel stride) and weight gradient (with a padded dy channel stride) against conv3d / autograd
    in fp32 on the same 16-bit operands"""
    g = torch.Generator().manual_seed(11)
    for (B, Cin, Cout, D, H, W, dtype) in ((1, 4, 48, 3, 8, 32, torch.bfloat16), (2, 1, 16, 2, 4, 64, torch.float16), (1, 3, 40, 2, 2, 128, torch.bfloat16)):
        x = torch.randn(B, Cin, D, H, W, generator=g).to(dtype)
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, 3, generator=g)).to(dtype)
        b = torch.randn(Cout, generator=g)
        assert ops_raw.stem_conv_supported(x, w)
        y = ops_raw.stem_conv_fwd(emu, x, w, b)
        wr = w.float().requires_grad_()
        ref = torch.nn.functional.conv3d(x.float(), wr, b, stride=1, padding=1)
        assert y.shape == ref.shape and (y.float() - ref).abs().max() <= (2e-2 if dtype == torch.bfloat16 else 4e-3) * max(1.0, float(ref.abs().max()))
        dy = torch.randn(ref.shape, generator=g).to(dtype)
        ref.backward(dy.float())
        x4 = ops_raw.stem_channel_last4(x)
        assert ops_raw.stem_wgrad_supported(x4, Cout, 3)
        dw = ops_raw.stem_conv_wgrad(emu, x4, dy, Cin, 3)
        assert dw.shape == wr.grad.shape and (dw - wr.grad).abs().max() <= 1e-3 * max(1.0, float(wr.grad.abs().max()))
        dyp = _channel_padded(dy, 64)                                         # dy as the instance-norm backward writes it at 128^3
        assert torch.equal(dw, ops_raw.stem_conv_wgrad(emu, x4, dyp, Cin, 3))


@pytest.mark.parametrize("B,Cin,Cout,D,H,W,dtype", [(1, 4, 48, 4, 8, 64, torch.bfloat16), (2, 1, 16, 2, 4, 128, torch.bfloat16),
                                                  (1, 3, 32, 6, 6, 64, torch.float16), (1, 2, 40, 2, 2, 256, torch.bfloat16)])
def test_stem_conv_wgrad_emulated(emu, B, Cin, Cout, D, H, W, dtype):
    """segm_stem_conv_wgrad (per output row a correlation along x on MFMA: A = dy row, B = stride-2 gathers of the staged
    channel-last-4 input row, per-slab partials added in order) against autograd of torch's conv3d in fp32 on the same 16-bit
    operands: 1, 2 and 4 k-steps per row, fewer than 4 input channels, channel counts that are not multiples of 16, volumes
    with every (kz, ky) tap touching the padding"""
    g = torch.Generator().manual_seed(B + Cin + Cout + D)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dtype)
    dy = torch.randn(B, Cout, D // 2, H // 2, W // 2, generator=g).to(dtype)
    x4 = ops_raw.stem_channel_last4(x)
    assert ops_raw.stem_wgrad_supported(x4, Cout)
    dw = ops_raw.stem_conv_wgrad(emu, x4, dy, Cin)
    w = torch.zeros(Cout, Cin, 7, 7, 7, requires_grad=True)
    torch.nn.functional.conv3d(x.float(), w, None, stride=2, padding=3).backward(dy.float())
    assert dw.shape == w.grad.shape and dw.dtype == torch.float32
    assert (dw - w.grad).abs().max() <= 1e-3 * max(1.0, float(w.grad.abs().max()))
    assert torch.equal(dw, ops_raw.stem_conv_wgrad(emu, x4, dy, Cin))                        # fixed summation order


@pytest.mark.parametrize("K,M,N,dtype,lda,ldb", [
    (4096, 192, 48, torch.bfloat16, None, None),        # in_proj at stage 0: three column blocks of a, aligned rows
    (1000, 35, 96, torch.bfloat16, None, None),         # x_proj: 35-column rows (2-byte loads for a), k tail (1000 = 31 * 32 + 8)
    (640, 96, 3, torch.float16, None, 35),              # dt_proj: b = the first 3 columns of the 35-column x_dbl
    (2048, 72, 200, torch.bfloat16, 80, 208),           # padded row strides, column blocks of b (96 + 96 + 8), m tail tile
    (96, 8, 8, torch.bfloat16, None, None),             # fewer chunks than waves
    (1000, 36, 96, torch.bfloat16, 40, None),           # round 6: x_dbl's padded layout (36 of 40 columns): whole 16-byte pieces that
    (999, 96, 38, torch.float16, None, 40),             # straddle `cols`, masked element loads in the matrix's last row; odd chunk counts
    (2080, 192, 48, torch.bfloat16, None, None),        # 65 chunks: the two-chunks-per-trip loop ends on its first half
])
def test_wgrad_gemm_tn_emulated(emu, K, M, N, dtype, lda, ldb):
    """segm_wgrad_gemm, layout TN (a^T b for token-major operands: the Mamba projections' weight gradients): LDS-staged 32-row
    tiles, fragments gathered column-wise, per-wave partials added in order - against the fp32 product of the same 16-bit
    operands; bitwise repeatable"""
    g = torch.Generator().manual_seed(K + M + N)
    A = torch.randn(K, lda or M, generator=g).to(dtype)
    B = torch.randn(K, ldb or N, generator=g).to(dtype)
    a, b = A[:, :M], B[:, :N]
    assert ops_raw.wgrad_gemm_tn_supported(a, b)
    out = ops_raw.wgrad_gemm(emu, a, b, ops_raw.WGEMM_TN)
    ref = a.float().t() @ b.float()
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert (out - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(out, ops_raw.wgrad_gemm(emu, a, b, ops_raw.WGEMM_TN))


@pytest.mark.parametrize("Bn,M,N,K,dtype", [(2, 48, 48, 2048, torch.bfloat16), (1, 4, 48, 512, torch.bfloat16), (2, 48, 4, 256, torch.float16),
                                           (1, 96, 96, 1024, torch.bfloat16), (3, 40, 20, 96, torch.bfloat16)])
def test_wgrad_gemm_nt_emulated(emu, Bn, M, N, K, dtype):
    """segm_wgrad_gemm, layout NT (sum over the batch of a[i] b[i]^T for channel-first volumes: the 1x1x1 convolutions' weight
    gradients): operand fragments straight from memory; channel slices of wider tensors as operands"""
    g = torch.Generator().manual_seed(Bn + M + N + K)
    A = torch.randn(Bn, M + 8, K, generator=g).to(dtype)
    Bm = torch.randn(Bn, N, K, generator=g).to(dtype)
    a = A[:, 8:]                                                            # a channel slice: batch stride != M * K
    assert ops_raw.wgrad_gemm_nt_supported(a, Bm)
    out = ops_raw.wgrad_gemm(emu, a, Bm, ops_raw.WGEMM_NT)
    ref = torch.einsum("bmk,bnk->mn", a.float(), Bm.float())
    assert out.shape == ref.shape and (out - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(out, ops_raw.wgrad_gemm(emu, a, Bm, ops_raw.WGEMM_NT))


def test_pointwise_weight_gradient_takes_the_library_gemm(emu, monkeypatch):
    """linear.pointwise's backward on channel-first activations: dW through segm_wgrad_gemm (layout NT) when the voxel count
    reaches the split threshold (lowered here), dx through segm_pointwise_cf - against autograd of the fp32 expression"""
    from segmamba_amd import lib as L, linear as LN
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(LN, "_MIN_K", 64)
    monkeypatch.setattr(LN, "_PW_MIN", 64)
    calls = []
    real = ops_raw.wgrad_gemm
    monkeypatch.setattr(ops_raw, "wgrad_gemm", lambda *a, **k: (calls.append(a[3]), real(*a, **k))[1])
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 48, 4, 4, 8, generator=g).bfloat16().requires_grad_()
    w = (0.2 * torch.randn(32, 48, generator=g)).requires_grad_()             # an fp32 master weight
    b = torch.randn(32, generator=g).requires_grad_()
    dy = torch.randn(2, 32, 4, 4, 8, generator=g).bfloat16()
    y = LN.pointwise(x, w, b)
    y.backward(dy)
    assert calls == [ops_raw.WGEMM_NT]
    xr = x.detach().float().requires_grad_()
    wr = w.detach().bfloat16().float().requires_grad_()
    br = b.detach().bfloat16().float().requires_grad_()
    ref = torch.einsum("oc,bcdhw->bodhw", wr, xr) + br.view(1, -1, 1, 1, 1)
    ref.backward(dy.float())
    assert (y.float() - ref).abs().max() <= 2e-2 * float(ref.abs().max())
    assert w.grad.dtype == torch.float32 and (w.grad - wr.grad).abs().max() <= 1e-4 * float(wr.grad.abs().max())
    assert b.grad.dtype == torch.float32 and (b.grad - br.grad).abs().max() <= 1e-4 * float(br.grad.abs().max())
    assert (x.grad.float() - xr.grad).abs().max() <= 2e-2 * float(xr.grad.abs().max())


def test_concatenated_input_convolution_is_one_node_with_in_place_parts(emu, monkeypatch):
    """conv3d_same_cat((a, b), w) - the decoder's conv1 on cat(upsampled, skip) - as ONE autograd node: the second part is
    added in place by the library kernel (every variant), gradients of both parts and of the whole weight, against autograd of
    conv3d on the materialised concatenation in fp32; and the route with an ordinary add (variant not tuned yet) gives the same"""
    from segmamba_amd import lib as L, conv3d as C3
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    monkeypatch.setattr(C3, "_CAT_FUSED", True)
    g = torch.Generator().manual_seed(8)
    a = torch.randn(1, 48, 2, 3, 64, generator=g).bfloat16().requires_grad_()
    b = torch.randn(1, 48, 2, 3, 64, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 96, 3, 3, 3, generator=g)).requires_grad_()             # fp32 master
    dy = torch.randn(1, 48, 2, 3, 64, generator=g).bfloat16()
    ar, br = a.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    wr = w.detach().bfloat16().float().requires_grad_()
    ref = torch.nn.functional.conv3d(torch.cat((ar, br), 1), wr, None, 1, 1)
    ref.backward(dy.float())
    outs = []
    for variant in (None,) + C3._HIP_VARIANTS:
        monkeypatch.setattr(C3, "_tuned_variant", lambda key, cands, variants, *rest, v=variant: v)
        for t in (a, b, w):
            t.grad = None
        y = C3.conv3d_same_cat((a, b), w)
        assert type(y.grad_fn).__name__ == "_ConvSameCatBackward"
        y.backward(dy)
        assert (y.float() - ref).abs().max() <= 2e-2 * float(ref.abs().max()), variant
        assert (a.grad.float() - ar.grad).abs().max() <= 2e-2 * float(ar.grad.abs().max())
        assert (b.grad.float() - br.grad).abs().max() <= 2e-2 * float(br.grad.abs().max())
        assert w.grad.dtype == torch.float32 and (w.grad - wr.grad).abs().max() <= 1e-3 * float(wr.grad.abs().max())
        outs.append(y.detach().float())
    for o in outs[1:]:
        assert (o - outs[0]).abs().max() <= 2e-2 * float(outs[0].abs().max())         # in place (fp32 sum, one rounding) vs add (two)
    monkeypatch.setattr(C3, "_CAT_FUSED", False)
    assert type(C3.conv3d_same_cat((a, b), w).grad_fn).__name__ == "AddBackward0"


def test_concatenated_input_pointwise_is_one_node_with_in_place_parts(emu, monkeypatch):
    """linear.pointwise_cat: the decoder's 1x1x1 residual convolution on cat(upsampled, skip) as one node, second part added in
    place by segm_pointwise_cf, against autograd of the fp32 expression on the materialised concatenation"""
    from segmamba_amd import lib as L, linear as LN
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(LN, "_PW_MIN", 64)
    monkeypatch.setattr(LN, "_MIN_K", 64)
    g = torch.Generator().manual_seed(12)
    a = torch.randn(2, 48, 2, 4, 16, generator=g).bfloat16().requires_grad_()
    b = torch.randn(2, 48, 2, 4, 16, generator=g).bfloat16().requires_grad_()
    w = (0.1 * torch.randn(48, 96, generator=g)).requires_grad_()
    bias = torch.randn(48, generator=g).requires_grad_()
    dy = torch.randn(2, 48, 2, 4, 16, generator=g).bfloat16()
    calls = []
    real = ops_raw.pointwise_cf
    monkeypatch.setattr(ops_raw, "pointwise_cf", lambda *args, **kw: (calls.append(bool(kw.get("accumulate"))), real(*args, **kw))[1])
    y = LN.pointwise_cat((a, b), w, bias)
    assert calls == [False, True] and type(y.grad_fn.next_functions[0][0]).__name__ == "_PointwiseCatBackward"   # behind the reshape
    y.backward(dy)
    ar, br = a.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    wr, biasr = w.detach().bfloat16().float().requires_grad_(), bias.detach().bfloat16().float().requires_grad_()
    ref = torch.einsum("oc,bcdhw->bodhw", wr, torch.cat((ar, br), 1)) + biasr.view(1, -1, 1, 1, 1)
    ref.backward(dy.float())
    assert (y.float() - ref).abs().max() <= 2e-2 * float(ref.abs().max())
    assert (a.grad.float() - ar.grad).abs().max() <= 2e-2 * float(ar.grad.abs().max())
    assert (b.grad.float() - br.grad).abs().max() <= 2e-2 * float(br.grad.abs().max())
    assert w.grad.dtype == torch.float32 and (w.grad - wr.grad).abs().max() <= 1e-3 * float(wr.grad.abs().max())
    assert (bias.grad - biasr.grad).abs().max() <= 1e-4 * float(biasr.grad.abs().max())


def test_decoder_block_with_fused_concatenation_emulated(emu, monkeypatch):
    """UnetResBlock(96 -> 48) on a (upsampled, skip) pair with SEGM_CONV_CAT_FUSED on (both cat convolutions as single nodes, parts
    added in place) and with the default route (per-part convolutions + adds), both against the block in fp32 on the same bf16
    inputs / weights.  bf16 rounding moves a few pre-activations across the LeakyReLU kink (single gradient entries change by
    O(1)), so the comparison is in the Frobenius norm: both routes sit ~4 % from the fp32 gradients, the fused one no further"""
    import torch.nn.functional as F
    from segmamba_amd import lib as L, unet_blocks as UB, conv3d as C3, linear as LN
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    monkeypatch.setattr(C3, "_tuned_variant", lambda key, cands, variants, *rest: variants[-1])
    monkeypatch.setattr(LN, "_PW_MIN", 64)
    torch.manual_seed(3)
    blk = UB.UnetResBlock(96, 48)
    g = torch.Generator().manual_seed(9)
    xa = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    xb = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    dy = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    w1, w2, w3 = (m.conv.weight.detach().bfloat16().float().requires_grad_() for m in (blk.conv1, blk.conv2, blk.conv3))
    ar, br = xa.float().requires_grad_(), xb.float().requires_grad_()
    x = torch.cat((ar, br), 1)
    o = F.instance_norm(F.conv3d(F.leaky_relu(F.instance_norm(F.conv3d(x, w1, None, 1, 1)), 0.01), w2, None, 1, 1))
    yr = F.leaky_relu(o + F.instance_norm(F.conv3d(x, w3)), 0.01)
    yr.backward(dy.float())
    refs = [yr.detach(), ar.grad, br.grad, w1.grad, w2.grad, w3.grad]
    errs = []
    for fused, front in ((False, False), (True, False), (True, True)):      # front: conv1 + conv3 as one node too (conv3d._ResFront)
        monkeypatch.setattr(C3, "_CAT_FUSED", fused)
        monkeypatch.setattr(C3, "_RES_FRONT", front)
        a, b = xa.clone().requires_grad_(), xb.clone().requires_grad_()
        blk.zero_grad()
        y = blk((a, b))
        y.backward(dy)
        got = [y.detach().float(), a.grad.float(), b.grad.float(), blk.conv1.conv.weight.grad, blk.conv2.conv.weight.grad, blk.conv3.conv.weight.grad]
        errs.append([float((u - r).norm() / r.norm()) for u, r in zip(got, refs)])
    for e0, e1, e2 in zip(*errs):
        assert e1 <= 8e-2 and e1 <= 1.25 * e0 + 1e-3, errs
        assert e2 <= 8e-2 and e2 <= 1.25 * e0 + 1e-3, errs
    assert errs[1][0] <= 1e-2 and errs[2][0] <= 1e-2            # the forward output itself: bf16 rounding only


@pytest.mark.parametrize("parts", [1, 2])
def test_res_front_node_matches_the_two_convolutions_emulated(emu, monkeypatch, parts):
    """conv3d.res_front: conv1 (3x3x3) and conv3 (1x1x1) of a residual block's input as one node.  Forward results and weight
    gradients are the two separate nodes' bit for bit (same launches in the same order); the data gradient of a part is ONE tensor -
    the 1x1x1 kernel writes it, the 3x3x3 kernel adds to it in place from its fp32 accumulators - where the two nodes round each
    contribution to bf16 and autograd adds them: one bf16 rounding apart, checked against the fp32 gradient."""
    import torch.nn.functional as F
    from segmamba_amd import lib as L, conv3d as C3, linear as LN
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    monkeypatch.setattr(C3, "_tuned_variant", lambda key, cands, variants, *rest: variants[-1])
    monkeypatch.setattr(C3, "_pick_was_hip", lambda *a, **k: True)
    monkeypatch.setattr(C3, "_CAT_FUSED", True)
    monkeypatch.setattr(LN, "_PW_MIN", 64)
    g = torch.Generator().manual_seed(31 + parts)
    shape = (1, 48, 2, 4, 64)
    xs = [torch.randn(shape, generator=g).bfloat16() for _ in range(parts)]
    cout = 48
    w1 = (0.05 * torch.randn(cout, 48 * parts, 3, 3, 3, generator=g)).bfloat16().float()
    w3 = (0.1 * torch.randn(cout, 48 * parts, 1, 1, 1, generator=g)).bfloat16().float()
    dy1 = torch.randn(1, cout, 2, 4, 64, generator=g).bfloat16()
    dy3 = torch.randn(1, cout, 2, 4, 64, generator=g).bfloat16()

    def run(front):
        ins = [x.clone().requires_grad_() for x in xs]
        a1, a3 = w1.clone().requires_grad_(), w3.clone().requires_grad_()
        if front:
            y1, st, y3 = C3.res_front(tuple(ins), a1, a3, want_stats=True)
            assert type(y1.grad_fn).__name__ == "_ResFrontBackward"
        elif parts == 1:
            (y1, st), y3 = C3.conv3d_same(ins[0], a1, None, want_stats=True), LN.pointwise(ins[0], a3.reshape(cout, -1))
        else:
            (y1, st), y3 = C3.conv3d_same_cat(tuple(ins), a1, want_stats=True), LN.pointwise_cat(tuple(ins), a3.reshape(cout, -1))
        torch.autograd.backward([y1, y3], [dy1, dy3])
        return y1.detach(), st, y3.detach(), a1.grad, a3.grad, [t.grad for t in ins]

    y1, st, y3, g1, g3, dxs = run(True)
    y1r, str_, y3r, g1r, g3r, dxr = run(False)
    assert torch.equal(y1, y1r) and torch.equal(y3.reshape(y3r.shape), y3r)
    assert st is not None and str_ is not None and torch.equal(st, str_)
    assert torch.equal(g1, g1r) and torch.equal(g3, g3r)
    # fp32 gradient of the parts
    ref = [x.float().requires_grad_() for x in xs]
    xc = torch.cat(ref, 1)
    torch.autograd.backward([F.conv3d(xc, w1, None, 1, 1), F.conv3d(xc, w3)], [dy1.float(), dy3.float()])
    for d, d2, r in zip(dxs, dxr, ref):
        scale = float(r.grad.abs().max())
        e1, e2 = float((d.float() - r.grad).abs().max()), float((d2.float() - r.grad).abs().max())
        assert e1 <= 2.0 ** -7 * scale and e1 <= 1.25 * e2 + 1e-6, (e1, e2, scale)
    monkeypatch.setattr(C3, "_RES_FRONT", False)
    assert C3.res_front(tuple(xs), w1, w3) is None


def test_every_routing_candidate_of_the_conv_dispatcher_runs_and_agrees(emu, monkeypatch):
    """conv3d.py's tuner runs EVERY candidate of a key once (timing) before it picks: here each candidate of the forward, the data
    gradient and the weight gradient of a 96 -> 48 layer is executed (vendor routes on the CPU, library routes on the emulator) and
    compared with the first one - what the GPU's first step does, minus the clock"""
    from segmamba_amd import lib as L, conv3d as C3
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    seen = {}

    def run_all(key, cands, *rest):
        outs = [c() for c in cands]
        seen[key[0]] = len(cands)
        for o in outs[1:]:
            assert o.shape == outs[0].shape
            assert (o.float() - outs[0].float()).abs().max() <= 3e-2 * max(1.0, float(outs[0].float().abs().max())), key[0]
        return outs[-1]

    monkeypatch.setattr(C3, "_pick", run_all)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 96, 2, 4, 16, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 96, 3, 3, 3, generator=g)).requires_grad_()                 # fp32 master
    b = torch.randn(48, generator=g).requires_grad_()
    y = C3.conv3d_same(x, w, b)
    y.backward(torch.randn(y.shape, generator=g).bfloat16())
    assert seen == {"fwd": 6, "dgrad": 7, "wgrad": 3}, seen       # vendor, blocked, four library variants (+ dgrad-as-forward)
    assert w.grad.dtype == torch.float32 and b.grad.dtype == torch.float32 and x.grad.dtype == torch.bfloat16
    # a layer the cube kernels take (96 -> 192 channels on an 8^3 volume): one more candidate each, the same results
    seen.clear()
    x2 = torch.randn(1, 96, 8, 8, 8, generator=g).bfloat16().requires_grad_()
    w2 = (0.05 * torch.randn(192, 96, 3, 3, 3, generator=g)).requires_grad_()
    y2, st = C3.conv3d_same(x2, w2, None, want_stats=True)
    y2.backward(torch.randn(y2.shape, generator=g).bfloat16())
    assert seen["fwd"] >= 2 and seen["dgrad"] >= 3 and seen["wgrad"] >= 2, seen
    assert st is None                    # (without a GPU the dispatcher's own pick is the vendor route: it hands no statistics on)
    assert w2.grad.dtype == torch.float32 and x2.grad.dtype == torch.bfloat16


@pytest.mark.parametrize("K,M,N,dtype,lda,ldb", [(5000, 96, 3, torch.bfloat16, 96, 40), (4096 + 77, 192, 6, torch.bfloat16, 192, 6),
                                                  (300, 72, 12, torch.float16, 80, 56), (9000, 384, 24, torch.bfloat16, 768, 64),
                                                  (1100, 768, 24, torch.bfloat16, 768, 56), (50, 2048, 5, torch.float16, 2048, 8)])
def test_skinny_tn_emulated(emu, K, M, N, dtype, lda, ldb):
    """segm_skinny_tn: wide^T skinny (the dt_proj weight gradient, reference selective_scan_interface.py:272) against an fp32
    matmul on the same rounded operands; views with row strides, ragged last slabs and passes, column tiles of 4 / 8 (three tiles for
    24 columns, narrowed by the fold buffer at 2048 channels); unsupported operands are refused"""
    g = torch.Generator().manual_seed(K + M)
    wide = torch.randn(K, lda, generator=g).to(dtype)[:, :M]
    skinny = torch.randn(K, ldb, generator=g).to(dtype)[:, 2:2 + N] if ldb > N + 2 else torch.randn(K, ldb, generator=g).to(dtype)[:, :N]
    out = ops_raw.skinny_tn(emu, wide, skinny)
    ref = wide.float().t() @ skinny.float()
    assert out.shape == (M, N) and out.dtype == torch.float32
    assert (out - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max()))
    assert not ops_raw.skinny_tn_supported(wide[:, :M - 2], skinny) and not ops_raw.skinny_tn_supported(wide[:, 1:], skinny[:, :N])
    with pytest.raises(RuntimeError):
        ops_raw.skinny_tn(emu, wide[:, :M - 2], skinny)


@pytest.mark.parametrize("K,M,N,ldb,off", [(5000, 96, 3, 40, 0), (3000, 192, 6, 40, 8), (700, 96, 24, 24, 0), (900, 96, 4, 4, 0)])
def test_skinny_tn_vector_rows_emulated(emu, monkeypatch, K, M, N, ldb, off):
    """segm_skinny_tn with the skinny operand's row segment aligned to its column tile (x_dbl in the padded layout: dt columns first):
    the tile's values come in ONE 8- / 16-byte load per row - the same sums, bit for bit, as the two-byte loads (SEGM_SKINNY_BVEC=0)"""
    g = torch.Generator().manual_seed(K + N)
    wide = torch.randn(K, M, generator=g).bfloat16()
    skinny = torch.randn(K, ldb, generator=g).bfloat16()[:, off:off + N]
    out = ops_raw.skinny_tn(emu, wide, skinny)
    monkeypatch.setenv("SEGM_SKINNY_BVEC", "0")
    out0 = ops_raw.skinny_tn(emu, wide, skinny)
    assert torch.equal(out, out0)
    ref = wide.float().t() @ skinny.float()
    assert (out - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape,dtype,pad", [((2, 48, 8, 16, 16), torch.bfloat16, 0), ((2, 6, 40, 40, 24), torch.bfloat16, 192),
                                             ((3, 5, 13), torch.float32, 0), ((1, 4, 70001), torch.float16, 0), ((2, 1, 9, 9, 9), torch.float32, 0)])
def test_channel_sum_emulated(emu, shape, dtype, pad):
    """segm_channel_sum (a convolution's bias gradient, `dy.sum((0, 2, 3, 4))`) against the fp64 sum of the same values: dense rows,
    rows with a padded channel stride, rows that start off a 16-byte boundary (scalar path), several segments per row"""
    g = torch.Generator().manual_seed(sum(shape))
    Bn, Cn = shape[:2]
    S = 1
    for n in shape[2:]:
        S *= n
    if pad:
        buf = torch.randn(Bn, Cn, S + pad, generator=g).to(dtype)
        x = buf[:, :, :S].unflatten(2, shape[2:])
    else:
        x = torch.randn(shape, generator=g).to(dtype)
    assert ops_raw.channel_sum_supported(x)
    out = ops_raw.channel_sum(emu, x)
    ref = x.double().sum([0] + list(range(2, x.dim())))
    assert out.shape == (Cn,) and out.dtype == torch.float32
    assert (out.double() - ref).abs().max() <= 1e-5 * max(1.0, float(x.double().abs().sum([0] + list(range(2, x.dim()))).max()))
    assert torch.equal(out, ops_raw.channel_sum(emu, x))
    if x.shape[1] > 1 and x.shape[2] > 1:
        assert not ops_raw.channel_sum_supported(x.transpose(1, 2))          # voxels no longer a unit-stride run


@pytest.mark.parametrize("dim,seqlen,chunk,order,ns,dtype,width", [
    (32, 128, 32, L.TIME_FORWARD, 1, torch.float32, 4), (64, 64, 32, L.TIME_REVERSED, 1, torch.bfloat16, 4),
    (16, 256, 64, L.TIME_INTERLEAVED, 8, torch.float32, 3), (96, 64, 32, L.TIME_INTERLEAVED, 8, torch.bfloat16, 4),
    (32, 64, 16, L.TIME_FORWARD, 1, torch.float16, 2)])
def test_scan_with_conv1d_inside_the_launch_emulated(emu, dim, seqlen, chunk, order, ns, dtype, width):
    """`conv_weight=`: the causal depthwise conv1d + SiLU formed inside the two scan passes from the conv INPUT (north star: "causal
    depthwise conv1d fused into the same launch") must equal segm_causal_conv1d_fwd followed by the scan BIT FOR BIT - output, gated
    output and the checkpoints the backward restarts from - in all three time orders (the halo in front of a chunk follows the
    order), for widths 2 - 4; a shape the regular kernels do not take is refused."""
    g = torch.Generator().manual_seed(dim + seqlen + width)
    Bn, N = 2, 16
    rn = lambda *s: torch.randn(*s, generator=g).to(dtype)
    x, z = rn(Bn, seqlen, dim), rn(Bn, seqlen, dim)
    delta = (0.5 * torch.rand(Bn, seqlen, dim, generator=g)).to(dtype)
    A = -0.5 * torch.rand(dim, N, generator=g)
    Bm, Cm = rn(Bn, seqlen, N), rn(Bn, seqlen, N)
    Dv, db = torch.randn(dim, generator=g), 0.5 * torch.rand(dim, generator=g)
    cw, cb = 0.5 * torch.randn(dim, width, generator=g), 0.1 * torch.randn(dim, generator=g)
    u = ops_raw.conv1d_fwd(emu, x, cw, cb, True, channel_last=True, time_order=order, nslices=ns)
    kw = dict(channel_last=True, time_order=order, nslices=ns, chunk=chunk, need_out=True, need_ckpt=True, need_last_state=True)
    ref = ops_raw.scan_fwd(emu, u, delta, A, Bm, Cm, Dv, z, db, True, **kw)
    assert ops_raw.scan_fused_conv_supported(emu, Bn, dim, seqlen, ns, order, chunk)
    fused = ops_raw.scan_fwd(emu, x, delta, A, Bm, Cm, Dv, z, db, True, conv_weight=cw, conv_bias=cb, **kw)
    for k in ("out", "out_z", "ckpt", "last_state"):
        assert torch.equal(fused[k], ref[k]), k
    inference = ops_raw.scan_fwd(emu, x, delta, A, Bm, Cm, Dv, z, db, True, conv_weight=cw, conv_bias=cb,
                                 **dict(kw, need_out=False, need_ckpt=False, need_last_state=False))
    assert inference["out"] is None and torch.equal(inference["out_z"], ref["out_z"])
    assert not ops_raw.scan_fused_conv_supported(emu, Bn, dim, seqlen - 3, ns, order, chunk)
    with pytest.raises(RuntimeError):                      # no gate: the kernels with the conv inside are built for softplus + gate
        ops_raw.scan_fwd(emu, x, delta, A, Bm, Cm, Dv, None, db, True, conv_weight=cw, conv_bias=cb, **kw)


def test_mamba_block_with_conv1d_inside_the_scan_launch_on_emulated_kernels(emu, monkeypatch):
    """SEGM_SCAN_FUSED_CONV1D=1: a Mamba(v3) block whose scan launches form u = SiLU(conv1d(x)) themselves == the default block bit
    for bit - output, input gradient, all 23 parameter gradients (the backward is the same code: it starts from the kept conv output)."""
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm import Mamba
    from tests.golden.make_golden import named_fill
    from segmamba_amd import selective_scan_interface as SSI
    m = Mamba(d_model=16, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=8)
    m.load_state_dict(named_fill(m.state_dict()))
    g = torch.Generator().manual_seed(4)
    x0, dy = torch.randn(2, 64, 16, generator=g), torch.randn(2, 64, 16, generator=g)
    seen, res = [], []
    real = ops_raw.scan_fwd_multi
    monkeypatch.setattr(ops_raw, "scan_fwd_multi", lambda lib, calls: (seen.append(sum("conv_weight" in c for c in calls)), real(lib, calls))[1])
    monkeypatch.setattr(SSI.L, "on_device", lambda t: True, raising=False)
    import segmamba_amd.mamba_simple as MS
    monkeypatch.setattr(MS.L, "on_device", lambda t: True, raising=False)
    for fused in (False, True):
        monkeypatch.setattr(SSI, "_FUSED_CONV1D", fused)
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        y = m(x)
        y.backward(dy)
        res.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert seen == [0, 3], seen                            # all three directions took the fused form
    (y0, gx0, gp0), (y1, gx1, gp1) = res
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    for k in gp0:
        assert torch.equal(gp0[k], gp1[k]), k


@pytest.mark.parametrize("dim,seqlen,chunk,order,ns,dtype,R,stride", [
    (32, 128, 32, L.TIME_FORWARD, 1, torch.float32, 3, 40), (64, 64, 32, L.TIME_REVERSED, 1, torch.bfloat16, 3, 40),
    (16, 256, 64, L.TIME_INTERLEAVED, 8, torch.float32, 6, 40), (96, 64, 32, L.TIME_INTERLEAVED, 8, torch.bfloat16, 8, 8),
    (32, 64, 16, L.TIME_FORWARD, 1, torch.float16, 1, 1), (128, 64, 32, L.TIME_REVERSED, 1, torch.bfloat16, 5, 5)])
def test_scan_with_dt_proj_inside_the_launch_emulated(emu, dim, seqlen, chunk, order, ns, dtype, R, stride):
    """`dt_x=, dt_weight=`: delta = dt_weight . x_dbl[:, :R] formed inside the two scan passes (reference
    selective_scan_interface.py:181-182 without its launch).  The delta tensor the apply pass writes must be the projection
    rounded to the element type (<= 1 ulp from the fp32-accumulated product of the same operands: the summation order differs),
    and the scan run on THAT delta must equal the fused launch bit for bit - outputs, checkpoints, last state - in all three time
    orders, for ranks 1 - 8 with tight and padded rows; a rank above 8 or an irregular shape is refused."""
    g = torch.Generator().manual_seed(dim + seqlen + R)
    Bn, N = 2, 16
    rn = lambda *s: torch.randn(*s, generator=g).to(dtype)
    u, z = rn(Bn, seqlen, dim), rn(Bn, seqlen, dim)
    rows = rn(Bn, seqlen, stride)
    dt_x = rows[:, :, :R]
    dt_w = (0.3 * torch.randn(dim, R, generator=g)).to(dtype).float().contiguous()
    A = -0.5 * torch.rand(dim, N, generator=g)
    Bm, Cm = rn(Bn, seqlen, N), rn(Bn, seqlen, N)
    Dv, db = torch.randn(dim, generator=g), 0.5 * torch.rand(dim, generator=g)
    kw = dict(channel_last=True, time_order=order, nslices=ns, chunk=chunk, need_out=True, need_ckpt=True, need_last_state=True)
    delta = torch.full((Bn, seqlen, dim), float("nan"), dtype=dtype)
    fused = ops_raw.scan_fwd(emu, u, delta, A, Bm, Cm, Dv, z, db, True, dt_x=dt_x, dt_weight=dt_w, **kw)
    exact = dt_x.double() @ dt_w.double().t()
    assert torch.isfinite(delta.float()).all()
    ulp = {torch.float32: 2.0 ** -23, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    assert ((delta.double() - exact).abs() <= ulp * exact.abs() + 1e-6).all()
    ref = ops_raw.scan_fwd(emu, u, delta.clone(), A, Bm, Cm, Dv, z, db, True, **kw)
    for k in ("out", "out_z", "ckpt", "last_state"):
        assert torch.equal(fused[k], ref[k]), k
    plain = ops_raw.scan_fwd(emu, u, torch.empty_like(delta), A, Bm, Cm, Dv, None, db, False, dt_x=dt_x, dt_weight=dt_w, **kw)
    ref2 = ops_raw.scan_fwd(emu, u, delta, A, Bm, Cm, Dv, None, db, False, **kw)        # the per-step-flag kernels
    assert torch.equal(plain["out"], ref2["out"]) and torch.equal(plain["ckpt"], ref2["ckpt"])
    with pytest.raises(RuntimeError):
        ops_raw.scan_fwd(emu, u, delta, A, Bm, Cm, Dv, z, db, True, dt_x=rn(Bn, seqlen, 9), dt_weight=torch.zeros(dim, 9), **kw)
    with pytest.raises(RuntimeError):                      # not a regular shape
        ops_raw.scan_fwd(emu, u[:, :-3], delta[:, :-3], A, Bm[:, :-3], Cm[:, :-3], Dv, z[:, :-3], db, True, dt_x=dt_x[:, :-3],
                         dt_weight=dt_w, **dict(kw, time_order=L.TIME_FORWARD, nslices=1))


def test_mamba_block_with_dt_proj_inside_the_scan_launch_on_emulated_kernels(emu, monkeypatch):
    """SEGM_SCAN_FUSED_DTPROJ=1: a bf16 Mamba(v3) block (row-streaming projections: the padded x_dbl layout the scan reads its dt
    rows from) whose scan launches form delta themselves, against the same block with the dt_proj launch - output, input
    gradient, all parameter gradients; the two deltas differ by at most an ulp of bf16 where the summation order rounds
    differently, so the comparison is at the bf16 level, not bit for bit."""
    monkeypatch.setattr(L, "_lib", emu)
    from mamba_ssm import Mamba
    from tests.golden.make_golden import named_fill
    from segmamba_amd import selective_scan_interface as SSI
    from segmamba_amd import linear as LN
    monkeypatch.setattr(LN, "_ROWS_HIP", True)
    monkeypatch.setattr(LN, "_ROWS_MIN", 1)
    monkeypatch.setattr(LN, "_on_device", lambda t: True)
    m = Mamba(d_model=16, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=8)
    m.load_state_dict(named_fill(m.state_dict()))
    m = m.bfloat16()
    g = torch.Generator().manual_seed(4)
    x0, dy = torch.randn(2, 64, 16, generator=g).bfloat16(), torch.randn(2, 64, 16, generator=g).bfloat16()
    seen, res = [], []
    real = ops_raw.scan_fwd_multi
    monkeypatch.setattr(ops_raw, "scan_fwd_multi", lambda lib, calls: (seen.append(sum("dt_x" in c for c in calls)), real(lib, calls))[1])
    monkeypatch.setattr(SSI.L, "on_device", lambda t: True, raising=False)
    import segmamba_amd.mamba_simple as MS
    monkeypatch.setattr(MS.L, "on_device", lambda t: True, raising=False)
    for fused in (False, True):
        monkeypatch.setattr(SSI, "_FUSED_DTPROJ", fused)
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        y = m(x)
        y.backward(dy)
        res.append((y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()}))
    assert seen == [0, 3], seen
    (y0, gx0, gp0), (y1, gx1, gp1) = res
    close = lambda a, b: float((a - b).abs().max()) <= 2e-2 * max(1e-2, float(b.abs().max()))
    assert close(y1, y0) and close(gx1, gx0)
    for k in gp0:
        assert close(gp1[k], gp0[k]), k


@pytest.mark.parametrize("shape,dtype,pad", [((2, 3, 2, 3, 16), torch.bfloat16, 0), ((1, 48, 1, 2, 8), torch.float16, 24), ((1, 2, 3, 1, 40), torch.bfloat16, 8)])
def test_depth_to_space2_emulated(emu, shape, dtype, pad):
    """segm_depth_to_space2: vol[b, c, 2z+i, 2y+j, 2x+k] = blk[b, c, i, j, k, z, y, x] and its inverse, bit for bit against the
    reshape / permute of fused_norm.patch_conv_transpose3d (reference unetr_block.py:52-60 does it inside ConvTranspose3d), into a
    dense volume and into one with a padded channel stride."""
    B, Cc, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    blk = torch.randn(B, Cc * 8, D, H_, W, generator=g).to(dtype)
    ref = blk.reshape(B, Cc, 2, 2, 2, D, H_, W).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, Cc, 2 * D, 2 * H_, 2 * W)
    out = None
    if pad:
        buf = torch.full((B, Cc, 8 * D * H_ * W + pad), float("nan"), dtype=dtype)
        out = buf[:, :, :8 * D * H_ * W].view(B, Cc, 2 * D, 2 * H_, 2 * W)
    vol = ops_raw.depth_to_space2(emu, blk, out=out)
    assert torch.equal(vol, ref)
    back = ops_raw.space_to_depth2(emu, vol)
    assert torch.equal(back, blk)
    with pytest.raises(RuntimeError):
        ops_raw.depth_to_space2(emu, blk[..., :W - 4].contiguous())          # W % 8


def test_transposed_conv_takes_the_depth_to_space_kernel(emu, monkeypatch):
    """patch_conv_transpose3d (kernel 2, stride 2) with the permute as the library kernel == the ATen permute path: output and
    the gradients of input, weight and bias, bit for bit (same GEMMs, same copies of the same values)."""
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    from segmamba_amd import fused_norm as FN
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(1, 16, 2, 2, 8, generator=g).bfloat16()
    w0 = (0.2 * torch.randn(16, 8, 2, 2, 2, generator=g)).bfloat16()
    b0 = torch.randn(8, generator=g).bfloat16()
    dy = torch.randn(1, 8, 4, 4, 16, generator=g).bfloat16()
    res = []
    for on in (False, True):
        monkeypatch.setattr(FN, "_D2S_HIP", on)
        x, w, b = x0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        y = FN.patch_conv_transpose3d(x, w, b, 2)
        y.backward(dy)
        res.append((y.detach(), x.grad, w.grad, b.grad))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)


def test_downsampling_conv_takes_the_space_to_depth_kernel(emu, monkeypatch):
    """patch_conv3d (kernel 2, stride 2: SegMamba's down-sampling convolutions, segmamba.py:145-150) with the gather as the library
    kernel + a channel-first GEMM == the permute path and == F.conv3d: output and the gradients of input, weight and bias."""
    monkeypatch.setattr(L, "_lib", emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    from segmamba_amd import fused_norm as FN
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(2, 8, 4, 6, 16, generator=g).bfloat16()
    w0 = (0.2 * torch.randn(16, 8, 2, 2, 2, generator=g)).bfloat16()
    b0 = torch.randn(16, generator=g).bfloat16()
    dy = torch.randn(2, 16, 2, 3, 8, generator=g).bfloat16()
    res = []
    for on in (False, True):
        monkeypatch.setattr(FN, "_D2S_HIP", on)
        x, w, b = x0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        y = FN.patch_conv3d(x, w, b, 2)
        y.backward(dy)
        res.append((y.detach().float(), x.grad.float(), w.grad.float(), b.grad.float()))
    xr, wr, br = x0.float().requires_grad_(), w0.float().requires_grad_(), b0.float().requires_grad_()
    yr = torch.nn.functional.conv3d(xr, wr, br, stride=2)
    yr.backward(dy.float())
    for got in res:
        for a, ref in zip(got, (yr.detach(), xr.grad, wr.grad, br.grad)):
            assert (a - ref).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("ipw", [2, 3, 5])
def test_conv3d_k3_wgrad_items_per_workgroup_emulated(emu, monkeypatch, ipw):
    """round 5: a workgroup accumulates `ipw` consecutive work items (batch / depth / x block / y part) before it writes its partial
    block - forced here on a small volume (the plan picks ipw > 1 only when there are more than 512 workgroups); also the v1 kernel."""
    g = torch.Generator().manual_seed(31 + ipw)
    x = torch.randn(2, 48, 3, 4, 128, generator=g).bfloat16()            # nxb = 2, items = 2 * 3 * 2 = 12
    dy = torch.randn(2, 96, 3, 4, 128, generator=g).bfloat16()
    ref = _wgrad_reference(x, dy)
    monkeypatch.setenv("SEGM_WGRAD_IPW", str(ipw))
    dw = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.float32)
    assert (dw - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-4
    monkeypatch.delenv("SEGM_WGRAD_IPW")
    monkeypatch.setenv("SEGM_WGRAD_V1", "1")
    dw1 = ops_raw.conv3d_k3_wgrad(emu, x, dy, torch.float32)
    assert (dw1 - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-4


@pytest.mark.parametrize("variant,shape", [("chain48", (2, 48, 48, 2, 5, 64)), ("chain48", (1, 96, 48, 2, 3, 72)), ("chain32", (1, 48, 96, 2, 4, 32)),
                                           ("chain32", (1, 96, 48, 1, 9, 40)),
                                           # H = 129 cut into 16 y parts of 9 rows: the last part is EMPTY and must still write a
                                           # zero-count slot (the partials buffer is torch.empty)
                                           ("chain48", (1, 48, 48, 1, 129, 16)), ("chain32", (1, 48, 48, 1, 129, 16))])
def test_conv3d_statistics_epilogue_feeds_instnorm_emulated(emu, variant, shape):
    """round 5: the chained 3x3x3 kernels sum {count, y, y^2} of what their storing K part writes (per workgroup and x pair); the
    InstanceNorm behind the convolution merges those partials instead of reading the volume again.  Same y as without the epilogue;
    the partials add up to the volume's sums; instnorm_fwd(stats=...) == instnorm_fwd() to rounding (the partials sum the fp32
    values before they are rounded to bf16)."""
    B, cin, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, cin, D, H_, W, generator=g).bfloat16()
    w = (0.1 * torch.randn(cout, cin, 3, 3, 3, generator=g)).bfloat16()
    kw = dict(chain=True, pitch48=True) if variant == "chain48" else dict(chain32=True)
    out = ref = stats = None
    blocks = list(range(0, cin, 48))
    for i, c0 in enumerate(blocks):
        wp = ops_raw.pack_conv3d_weight(w[:, c0:c0 + 48])
        ref = ops_raw.conv3d_k3_fwd(emu, x[:, c0:c0 + 48], wp, None, out=ref, accumulate=i > 0, **kw)
        if i + 1 < len(blocks):
            out = ops_raw.conv3d_k3_fwd(emu, x[:, c0:c0 + 48], wp, None, out=out, accumulate=i > 0, **kw)
        else:
            out, stats = ops_raw.conv3d_k3_fwd(emu, x[:, c0:c0 + 48], wp, None, out=out, accumulate=i > 0, want_stats=True, **kw)
    assert torch.equal(out, ref) and stats is not None and stats.shape[:2] == (B, cout) and stats.shape[3] == 4
    yf = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    assert torch.equal(stats[..., 0].sum(-1), torch.full((B, cout), float(D * H_ * W)))
    # later 48-channel blocks add to the ROUNDED result of the earlier ones: the sums are those of the stored values up to that rounding
    tol = 2e-2 if len(blocks) > 1 else 2e-5
    assert (stats[..., 1].sum(-1) - yf.sum((2, 3, 4))).abs().max() <= tol * yf.abs().sum((2, 3, 4)).max()
    assert (stats[..., 2].sum(-1) - (yf * yf).sum((2, 3, 4))).abs().max() <= tol * (yf * yf).sum((2, 3, 4)).max()
    y0, m0, r0 = ops_raw.instnorm_fwd(emu, out, None, "leaky_relu", 0.01, 1e-5)
    y1, m1, r1 = ops_raw.instnorm_fwd(emu, out, None, "leaky_relu", 0.01, 1e-5, stats=stats)
    assert (m0 - m1).abs().max() <= 3e-3 * out.float().abs().max() and (r0 / r1 - 1).abs().max() <= 3e-3
    assert (y0.float() - y1.float()).abs().max() <= 3e-2


@pytest.mark.parametrize("cat_fused", [False, True])
def test_unet_res_block_with_statistics_from_the_convolutions_emulated(emu, monkeypatch, cat_fused):
    """round 5: UnetResBlock forward + backward with the InstanceNorm statistics summed in the convolutions' epilogues (conv3d._STATS,
    the default) against the same block with every InstanceNorm making its own pass: outputs and all gradients agree to the
    rounding of the statistics (fp32 sums before vs after the bf16 rounding of the activations), and the statistics launches
    really are skipped (two of the block's three InstanceNorms follow a 3x3x3 convolution; with the cat convolution as one node -
    SEGM_CONV_CAT_FUSED - the one behind the concatenated input too)."""
    from segmamba_amd import lib as L, unet_blocks as UB, conv3d as C3
    monkeypatch.setattr(L, "get_lib", lambda: emu)
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())     # the library's candidates: chain32 is the last variant
    monkeypatch.setattr(C3, "_pick_was_hip", lambda *a: True)
    monkeypatch.setattr(C3, "_tuned_variant", lambda key, cands, variants, width=0: variants[-1])
    monkeypatch.setattr(C3, "_CAT_FUSED", cat_fused)
    torch.manual_seed(3)
    blk = UB.UnetResBlock(96, 48).bfloat16()
    g = torch.Generator().manual_seed(9)
    xa = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    xb = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    dy = torch.randn(1, 48, 2, 4, 64, generator=g).bfloat16()
    res, used = [], []
    real = ops_raw.instnorm_fwd
    monkeypatch.setattr(ops_raw, "instnorm_fwd", lambda *a, **k: (used.append(k.get("stats") is not None), real(*a, **k))[1])
    for stats in (False, True):
        monkeypatch.setattr(C3, "_STATS", stats)
        used.clear()
        a, b = xa.clone().requires_grad_(), xb.clone().requires_grad_()
        blk.zero_grad()
        y = blk((a, b))
        y.backward(dy)
        res.append([y.detach().clone(), a.grad.clone(), b.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
        assert used == ([False, False, False] if not stats else ([True, False, True] if cat_fused else [False, False, True])), used
    for u, v in zip(*res):
        # a pre-activation that sits at the LeakyReLU kink may change sides with the last bit of the statistics: compare in the bulk
        # (each such flip changes that voxel's gradient a hundredfold, and the data gradient spreads it over 27 neighbours: in this
        # 512-voxel volume a handful of flips are ~1 % of a gradient tensor; the forward output must agree everywhere)
        bad = ((u.float() - v.float()).abs() > 2e-2 * max(1.0, float(u.float().abs().max()))).float().mean()
        assert bad <= (0.0 if u is res[0][0] else 2e-2), float(bad)


# ---- channel-last 3x3x3 convolution (round 6 prototype, csrc/conv3d_cl.hip) ---------------------------------------------------------
@pytest.mark.parametrize("shape,waves8", [((1, 2, 3, 16), False), ((1, 1, 2, 32), False), ((2, 2, 2, 64), False), ((1, 1, 1, 128), False),
                                          ((1, 2, 2, 32), True), ((1, 1, 2, 64), True), ((1, 1, 3, 16), True)])
def test_conv3d_k3_fwd_channel_last_emulated(emu, shape, waves8):
    """kx taps in N + shift-and-add of accumulators along x (DPP row shifts, the tile that waits for the next group's first voxel),
    zero padding by buffer range check (rows outside the volume) and by absence (x), the weight image in LDS, the channel deal
    of the three co tiles, bias, in-place accumulation; every group width (one, two, four tiles), one and two groups per row."""
    B, D, H_, W = shape
    g = torch.Generator().manual_seed(sum(shape) + (7 if waves8 else 0))
    x = torch.randn(B, 48, D, H_, W, generator=g).bfloat16()
    w = (0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
    bias = torch.randn(48, generator=g)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    tol = 1e-2 * max(1.0, float(ref.abs().max()))
    xcl = x.permute(0, 2, 3, 4, 1).contiguous()
    img = ops_raw.conv3d_cl_weight_image(emu, w)
    y = ops_raw.conv3d_k3_fwd_cl(emu, xcl, img, bias, waves8=waves8)
    assert y.shape == (B, D, H_, W, 48) and y.dtype == torch.bfloat16
    assert (y.permute(0, 4, 1, 2, 3).float() - ref).abs().max() <= tol
    # a second 48-channel input block accumulated in place, on padded (strided) volumes
    x2 = torch.randn(B, 48, D, H_, W, generator=g).bfloat16()
    w2 = (0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)).bfloat16()
    ref2 = ref + torch.nn.functional.conv3d(x2.float(), w2.float(), None, 1, 1)
    pad = torch.zeros(B, D, H_, W, 56, dtype=torch.bfloat16)
    pad[..., :48] = x2.permute(0, 2, 3, 4, 1)
    ya = torch.full((B, D, H_, W, 64), 7.0, dtype=torch.bfloat16)
    ya[..., :48] = y
    ops_raw.conv3d_k3_fwd_cl(emu, pad[..., :48], ops_raw.conv3d_cl_weight_image(emu, w2), None, out=ya[..., :48], accumulate=True, waves8=waves8)
    assert (ya[..., :48].permute(0, 4, 1, 2, 3).float() - ref2).abs().max() <= 2 * tol
    assert bool((ya[..., 48:] == 7.0).all())              # nothing written beyond the 48 channels of a voxel


def test_conv3d_k3_dgrad_channel_last_emulated(emu):
    """the data gradient is the same kernel on dy with the image of flip(w).transpose(0, 1)"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 48, 2, 3, 32, generator=g, requires_grad=True)
    w = 0.1 * torch.randn(48, 48, 3, 3, 3, generator=g)
    dy = torch.randn(1, 48, 2, 3, 32, generator=g).bfloat16()
    torch.nn.functional.conv3d(x, w.bfloat16().float(), None, 1, 1).backward(dy.float())
    wt = torch.flip(w, (2, 3, 4)).transpose(0, 1).contiguous().bfloat16()
    dx = ops_raw.conv3d_k3_fwd_cl(emu, dy.permute(0, 2, 3, 4, 1).contiguous(), ops_raw.conv3d_cl_weight_image(emu, wt))
    assert (dx.permute(0, 4, 1, 2, 3).float() - x.grad).abs().max() <= 1e-2 * max(1.0, float(x.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_add3_emulated(emu, dtype):
    """segm_add3: a + b + c in one pass (fp32 sum, one rounding), also in place on a; packet counts that are odd / below one
    grid stride (the two-packets-in-flight loop)"""
    g = torch.Generator().manual_seed(3)
    for n in (8, 8 * 257, 8 * 4096 + 24):
        a, b, c = (torch.randn(n, generator=g).to(dtype) for _ in range(3))
        ref = (a.float() + b.float() + c.float()).to(dtype)
        out = ops_raw.add3(emu, a, b, c)
        assert torch.equal(out, ref)
        a2 = a.clone()
        assert ops_raw.add3(emu, a2, b, c, out=a2) is a2 and torch.equal(a2, ref)
    with pytest.raises(RuntimeError):
        ops_raw.add3(emu, a[:10], b[:10], c[:10])           # not whole 16-byte packets


@pytest.mark.parametrize("shape,nt,splits", [((2, 64, 64, 8, 8, 8), 2, 2), ((1, 96, 128, 8, 16, 8), 4, 3), ((1, 32, 96, 16, 8, 16), 3, 1),
                                             ((1, 64, 192, 8, 8, 16), 0, 0)])
def test_conv3d_k3_cube_forward_emulated(emu, shape, nt, splits):
    """segm_conv3d_k3_cube_fwd (ABI 10): 8 x 8 x 8 cubes x 64 / 96 / 128 output channels per workgroup, the halo cube of 32 input
    channels per round through LDS, split contraction + fixed-order reduction; against fp32 ATen on the 16-bit inputs.  Volumes of
    one cube (all halos are padding) and of several (halos from the neighbour cube, left / right x halo loads), bias, and a second
    input part accumulated in place on strided tensors"""
    B, cin, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, D, H_, W, generator=g).bfloat16()
    w = (0.1 * torch.randn(cout, cin, 3, 3, 3, generator=g)).bfloat16()
    bias = torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    tol = 1e-2 * max(1.0, float(ref.abs().max()))
    img = ops_raw.conv3d_cube_weight_image(emu, w)
    pnt, ps, need = ops_raw.conv3d_cube_plan(emu, B, cin, cout, D, H_, W, nt, splits)
    assert (nt == 0 or pnt == nt) and (splits == 0 or ps == splits) and need == (0 if ps == 1 else ps * B * cout * D * H_ * W)
    y = ops_raw.conv3d_k3_cube_fwd(emu, x, img, cout, bias, nt=nt, splits=splits)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    assert (y.float() - ref).abs().max() <= tol
    # the same launch with another plan: the same sums in another association
    y1, st = ops_raw.conv3d_k3_cube_fwd(emu, x, img, cout, bias, nt=2 if cout % 64 == 0 else 3, splits=cin // 32, want_stats=True)
    assert (y1.float() - ref).abs().max() <= tol
    # the InstanceNorm partials of that launch: per (batch, channel) 512-voxel parts of {count, sum, sum of squares}
    # (512 voxels per part from the reduction launch; 128 - one wave's two z planes of a cube - when the single split stores directly)
    vol = D * H_ * W
    chunk = 128 if cin // 32 == 1 else 512
    assert st.shape == (B, cout, vol // chunk, 4) and bool((st[..., 0] == chunk).all())
    tot = ref.double().reshape(B, cout, vol)
    assert (st[..., 1].double().sum(-1) - tot.sum(-1)).abs().max() <= vol * tol
    assert (st[..., 2].double().sum(-1) - (tot * tot).sum(-1)).abs().max() <= 1e-2 * float((tot * tot).sum(-1).max())
    if chunk == 512:
        r64 = tot.reshape(B, cout, vol // 512, 512)
        assert (st[..., 1].double() - r64.sum(-1)).abs().max() <= 512 * tol
    yn, m_, r_ = ops_raw.instnorm_fwd(emu, y1, None, "none", stats=st)
    yn0, m0, r0 = ops_raw.instnorm_fwd(emu, y1, None, "none")
    assert (m_ - m0).abs().max() <= 2e-2 * max(1.0, float(m0.abs().max())) and (r_ / r0 - 1).abs().max() <= 2e-2
    # a second part of the input accumulated in place; input and output with padded channel / batch strides
    x2 = torch.randn(B, cin, D, H_, W, generator=g).bfloat16()
    w2 = (0.1 * torch.randn(cout, cin, 3, 3, 3, generator=g)).bfloat16()
    ref2 = ref + torch.nn.functional.conv3d(x2.float(), w2.float(), None, 1, 1)
    xp = torch.zeros(B, cin + 1, D, H_, W + 8, dtype=torch.bfloat16)[:, :cin, :, :, :W]
    xp.copy_(x2)
    yp = torch.full((B, cout + 2, D, H_, W + 8), 7.0, dtype=torch.bfloat16)
    yv = yp[:, :cout, :, :, :W]
    yv.copy_(y)
    ops_raw.conv3d_k3_cube_fwd(emu, xp, ops_raw.conv3d_cube_weight_image(emu, w2), cout, None, out=yv, accumulate=True, nt=nt, splits=splits)
    assert (yv.float() - ref2).abs().max() <= 2 * tol
    assert bool((yp[:, cout:] == 7.0).all()) and bool((yp[:, :, :, :, W:] == 7.0).all())      # nothing written outside the view


def test_conv3d_k3_cube_dgrad_and_errors_emulated(emu):
    """the data gradient is the same launch on dy with the flipped image (Cout = the weight's Cin); argument errors"""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 64, 8, 8, 16, generator=g, requires_grad=True)
    w = (0.1 * torch.randn(96, 64, 3, 3, 3, generator=g)).bfloat16()
    dy = torch.randn(1, 96, 8, 8, 16, generator=g).bfloat16()
    torch.nn.functional.conv3d(x, w.float(), None, 1, 1).backward(dy.float())
    dx = ops_raw.conv3d_k3_cube_fwd(emu, dy, ops_raw.conv3d_cube_weight_image(emu, w, flipped=True), 64)
    assert (dx.float() - x.grad).abs().max() <= 1e-2 * max(1.0, float(x.grad.abs().max()))
    img = ops_raw.conv3d_cube_weight_image(emu, w)
    assert not ops_raw.conv3d_cube_supported(dy[:, :, :, :, :12], 64)              # width not a multiple of 8
    assert not ops_raw.conv3d_cube_supported(dy[:, :40], 64)                        # Cin not a multiple of 32
    assert not ops_raw.conv3d_cube_supported(dy, 160)                               # Cout neither a multiple of 64 nor of 96
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_cube_plan(emu, 1, 64, 96, 8, 8, 12)
    xs = x.detach().bfloat16()
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_k3_cube_fwd(emu, xs, img, 96, nt=4)                          # 96 channels cannot run 128-channel blocks
    with pytest.raises(RuntimeError):
        ops_raw.conv3d_k3_cube_fwd(emu, xs, img, 96, splits=3)                      # only two rounds to split


@pytest.mark.parametrize("shape", [(2, 32, 64, 8, 8, 16), (1, 64, 128, 8, 16, 24), (3, 32, 64, 16, 8, 8)])
def test_conv3d_k3_cube_wgrad_emulated(emu, shape):
    """segm_conv3d_k3_cube_wgrad: 64 x 32 (co, ci) channels x 27 taps per workgroup, dY cube and X halo cube in LDS in their row
    layout, the kx = 0 / 2 operands by register shifts with the neighbour elements (interior octets of wider rows: left and right
    neighbours; 8-wide rows: padding), cube ranges split over workgroups + fixed-order reduction; against autograd on the 16-bit
    operands, fp32 and 16-bit results, strided inputs"""
    B, cin, cout, D, H_, W = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, cin, D, H_, W, generator=g).bfloat16()
    dy = torch.randn(B, cout, D, H_, W, generator=g).bfloat16()
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    torch.nn.functional.conv3d(x.float(), w, None, 1, 1).backward(dy.float())
    ref = w.grad
    tol = 1e-2 * float(ref.abs().max())
    assert ops_raw.conv3d_cube_wgrad_supported(x, dy)
    dw = ops_raw.conv3d_k3_cube_wgrad(emu, x, dy, torch.float32)
    assert dw.shape == ref.shape and (dw - ref).abs().max() <= 1e-3 * float(ref.abs().max())        # fp32 accumulation of exact products
    dwb = ops_raw.conv3d_k3_cube_wgrad(emu, x, dy, torch.bfloat16)
    assert dwb.dtype == torch.bfloat16 and (dwb.float() - ref).abs().max() <= tol
    xp = torch.zeros(B, cin + 3, D, H_, W + 8, dtype=torch.bfloat16)[:, :cin, :, :, :W]
    xp.copy_(x)
    dyp = torch.zeros(B + 1, cout, D, H_, W + 16, dtype=torch.bfloat16)[:B, :, :, :, :W]
    dyp.copy_(dy)
    assert torch.equal(ops_raw.conv3d_k3_cube_wgrad(emu, xp, dyp, torch.float32), dw)
    assert not ops_raw.conv3d_cube_wgrad_supported(x[:, :, :, :, :4], dy[:, :, :, :, :4])
    assert not ops_raw.conv3d_cube_wgrad_supported(x[:, :16], dy)


def test_gather16_emulated(emu):
    """segm_gather16: out[i] = src[map(i)], an index per element (mode 0) or (first index, step) per eight elements (mode 1); the
    compact form exists exactly when every group of eight is an arithmetic progression - e.g. the cube kernels' weight images"""
    g = torch.Generator().manual_seed(2)
    src = torch.randn(5000, generator=g).bfloat16()
    idx = torch.randint(0, 5000, (8 * 333,), generator=g, dtype=torch.int32)
    out = torch.empty(idx.numel(), dtype=torch.bfloat16)
    assert torch.equal(ops_raw.gather16(emu, src, idx, out), src[idx.long()])
    assert ops_raw.gather16_compact_map(idx) is None
    w = torch.randn(64, 32, 3, 3, 3, generator=g).bfloat16()
    for flipped in (False, True):
        m = ops_raw.conv3d_cube_index(emu, 64, 32, flipped, "cpu")
        cm = ops_raw.gather16_compact_map(m)
        assert cm is not None and cm.shape == (m.numel() // 8, 2)
        out = torch.empty(m.numel(), dtype=torch.bfloat16)
        assert torch.equal(ops_raw.gather16(emu, w.reshape(-1), cm, out, compact=True), w.reshape(-1)[m.long()])
    with pytest.raises(RuntimeError):
        ops_raw.gather16(emu, src, idx[:12], torch.empty(12, dtype=torch.bfloat16))


def test_conv3d_cube_pack_multi_emulated(emu):
    """segm_conv3d_k3_cube_pack_multi: the fragment images of several weights (whole weights and a channel slice of a wider one, forward
    and data-gradient images) from one 16-bit buffer in one launch == the images the exported index map describes, element for element"""
    g = torch.Generator().manual_seed(8)
    buf = torch.randn(8 + 64 * 32 * 27 + 64 * 96 * 27 + 16, generator=g).bfloat16()
    w1 = buf[8:8 + 64 * 32 * 27].view(64, 32, 3, 3, 3)
    wide = buf[8 + 64 * 32 * 27:8 + 64 * 32 * 27 + 64 * 96 * 27].view(64, 96, 3, 3, 3)
    w2 = wide[:, 32:96]                                     # a cat part: (64, 64) with the wide weight's channel stride
    items, expect, o = [], [], 0
    for w, fl in ((w1, False), (w1, True), (w2, False), (w2, True)):
        off = (w.data_ptr() - buf.data_ptr()) // 2
        items.append((off, o, w.shape[0], w.shape[1], w.stride(0), fl))
        expect.append(ops_raw.conv3d_cube_weight_image(emu, w, fl, by_index=True))
        o += w.shape[0] * w.shape[1] * 27
    descs, nblocks = ops_raw.cube_pack_descs(items, "cpu")
    assert nblocks == sum((w.shape[0] // 16) * (w.shape[1] // 32) if not fl else (w.shape[1] // 16) * (w.shape[0] // 32)
                          for w, fl in ((w1, False), (w1, True), (w2, False), (w2, True)))
    out = torch.zeros(o, dtype=torch.bfloat16)
    ops_raw.conv3d_cube_pack_multi(emu, buf, out, descs, nblocks)
    assert torch.equal(out, torch.cat(expect))
