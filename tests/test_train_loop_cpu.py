"""Host-side pieces of the training loop that need no GPU: the reference's fp16 + GradScaler step and the device-side
augmentation feeder."""
import torch
import torch.nn as nn

from segmamba_amd.augment import DeviceAugmenter
from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step


def _tiny():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv3d(4, 8, 3, padding=1), nn.InstanceNorm3d(8), nn.LeakyReLU(0.01), nn.Conv3d(8, 4, 1))


def test_fp16_gradscaler_step_follows_the_reference_loop():
    """amp='fp16': scale(loss).backward -> unscale_ -> clip -> scaler.step -> scaler.update (light_training/trainer.py:461-466).
    On CPU tensors (torch's CPU autocast / GradScaler): the parameters move, the scale is the GradScaler default, and a step whose
    gradients overflow is skipped with the scale halved."""
    st = build_training_state(torch.device("cpu"), model=_tiny(), amp="fp16")
    assert st.autocast_dtype == torch.float16 and st.scaler is not None and st.scaler.get_scale() == 65536.0
    g = torch.Generator().manual_seed(1)
    img, lab = torch.rand(1, 4, 8, 8, 8, generator=g), torch.randint(0, 4, (1, 8, 8, 8), generator=g)
    before = [p.detach().clone() for p in st.model.parameters()]
    loss = train_step(st, img, lab)
    assert torch.isfinite(loss)
    assert any((a - b.detach()).abs().max() > 0 for a, b in zip(before, st.model.parameters()))
    assert st.step == 1 and st.scaler.get_scale() == 65536.0
    # an overflowing step: inf in the input -> non-finite gradients -> optimizer step skipped, scale halved
    before = [p.detach().clone() for p in st.model.parameters()]
    bad = img.clone()
    bad[0, 0, 0, 0, 0] = float("inf")
    train_step(st, bad, lab)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, st.model.parameters()))
    assert st.scaler.get_scale() == 32768.0


def test_bf16_is_the_default_and_has_no_scaler():
    st = build_training_state(torch.device("cpu"), model=_tiny())
    assert st.autocast_dtype == torch.bfloat16 and st.scaler is None


class _OnlyMirror(DeviceAugmenter):
    def _coin(self, p, *shape):                            # every p = 0.5 decision is "yes" (the mirrors), every other one "no"
        return torch.full(shape, p == 0.5, dtype=torch.bool)


def test_augmenter_mirrors_image_and_label_together():
    g = torch.Generator().manual_seed(0)
    lab = torch.randint(0, 4, (2, 6, 5, 4), generator=g)
    img = torch.stack([lab.float(), lab.float() * 2, torch.rand(2, 6, 5, 4, generator=g), torch.zeros(2, 6, 5, 4)], 1)
    x, y = _OnlyMirror("cpu")(img, lab)
    assert torch.equal(y, lab.flip(1, 2, 3)) and torch.equal(x, img.flip(2, 3, 4))
    assert torch.equal(x[:, 0], y.float())                 # still aligned


def test_augmenter_is_reproducible_and_keeps_shapes_and_classes():
    g = torch.Generator().manual_seed(3)
    img, lab = torch.rand(2, 4, 12, 12, 12, generator=g), torch.randint(0, 4, (2, 12, 12, 12), generator=g)
    outs = []
    for _ in range(2):
        aug = DeviceAugmenter("cpu", seed=7)
        outs.append([aug(img, lab) for _ in range(12)])     # 12 draws: every transform fires at least once at these odds
    for (x0, y0), (x1, y1) in zip(*outs):
        assert torch.equal(x0, x1) and torch.equal(y0, y1)
        assert x0.shape == img.shape and y0.shape == lab.shape and y0.dtype == lab.dtype and x0.dtype == img.dtype
        assert torch.isfinite(x0).all() and int(y0.min()) >= 0 and int(y0.max()) <= 3
    assert any(not torch.equal(x, img) for x, _ in outs[0])


def test_intensity_transforms_follow_their_published_definitions():
    aug = DeviceAugmenter("cpu", seed=0, spatial=False)
    x = torch.rand(3, 2, 8, 8, 8, generator=torch.Generator().manual_seed(5)) * 3 - 1
    # gamma with retained statistics: per-channel mean and std unchanged wherever it was applied
    class G(DeviceAugmenter):
        def _coin(self, p, *shape):
            return torch.ones(shape, dtype=torch.bool)
    y = G("cpu", seed=1, spatial=False)._gamma(x, 0.3, invert=False)
    assert torch.allclose(y.mean((2, 3, 4)), x.mean((2, 3, 4)), atol=1e-5)
    assert torch.allclose(y.std((2, 3, 4)), x.std((2, 3, 4)), rtol=1e-4)
    yi = G("cpu", seed=1, spatial=False)._gamma(x, 0.1, invert=True)
    assert torch.allclose(yi.mean((2, 3, 4)), x.mean((2, 3, 4)), atol=1e-5)
    # blur of a constant volume is the constant; low-res simulation keeps the value range
    c = torch.full((1, 2, 8, 8, 8), 2.5)
    assert torch.allclose(G("cpu", seed=2, spatial=False)._blur(c), c, atol=1e-6)
    lo = G("cpu", seed=3, spatial=False)._low_res(x)
    assert float(lo.min()) >= float(x.min()) - 1e-6 and float(lo.max()) <= float(x.max()) + 1e-6
    assert aug is not None


def test_synthetic_feeder_with_augmentation():
    data = SyntheticBraTS(1, 8, torch.device("cpu"), seed=42, augment=True)
    a, b = data.next(), data.next()
    assert a[0].shape == (1, 4, 8, 8, 8) and a[1].shape == (1, 8, 8, 8) and b[0].shape == a[0].shape


def test_param_bank_hands_out_views_of_one_converted_buffer():
    """param_bank.ParamBank: the parameters become views of one flat fp32 buffer (state_dict unchanged); inside `bank.step()`
    `low_precision` returns pieces of ONE 16-bit buffer converted on entry - for a parameter and for any view of it (reshaped,
    transposed, channel-sliced weights) - without an autograd edge; outside it is a plain cast; after an in-place parameter
    update the next step sees the new values"""
    from segmamba_amd.param_bank import ParamBank, low_precision
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(5, 3), nn.Conv3d(4, 6, 3))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    bank = ParamBank(m, torch.bfloat16)
    assert list(m.state_dict()) == list(before) and all(torch.equal(m.state_dict()[k], v) for k, v in before.items())
    w = m[1].weight                                            # NOT the first parameter of the flat buffer
    assert w.is_leaf and w.requires_grad and w.data_ptr() == bank.flat32.data_ptr() + 4 * (16 + 8)
    plain = low_precision(w, torch.bfloat16)
    assert plain.dtype == torch.bfloat16 and not plain.requires_grad
    assert not bank.flat16.data_ptr() <= plain.data_ptr() < bank.flat16.data_ptr() + 2 * bank.flat16.numel()
    with bank.step():
        a = low_precision(w, torch.bfloat16)
        assert a.data_ptr() == bank.flat16.data_ptr() + 2 * (16 + 8) and not a.requires_grad and torch.equal(a, w.detach().bfloat16())
        for view in (w.reshape(6, -1), w.reshape(6, -1).t(), w[:, 1:3], w[2:4, :, 1]):
            v = low_precision(view, torch.bfloat16)
            assert torch.equal(v, view.detach().bfloat16())
            assert bank.flat16.data_ptr() <= v.data_ptr() < bank.flat16.data_ptr() + 2 * bank.flat16.numel()       # no copy
        f = low_precision(w.flip(2), torch.bfloat16)                     # not a view of the parameter: a plain cast
        assert torch.equal(f, w.detach().flip(2).bfloat16())
        assert torch.equal(low_precision(m[0].bias, torch.bfloat16), m[0].bias.detach().bfloat16())
        assert low_precision(w, torch.float16).dtype == torch.float16    # not the bank's dtype: a plain cast
        assert low_precision(None, torch.bfloat16) is None
    with torch.no_grad():
        w.mul_(2.0)
    stale = low_precision(w, torch.bfloat16)                                         # stale copies are not handed out
    assert not bank.flat16.data_ptr() <= stale.data_ptr() < bank.flat16.data_ptr() + 2 * bank.flat16.numel()
    with bank.step():
        assert torch.equal(low_precision(w, torch.bfloat16), w.detach().bfloat16())


def test_param_bank_declines_parameters_whose_storage_moved():
    """anything that re-points `p.data` after the bank was built (model.to(memory_format=...), .to(dtype),
    load_state_dict(assign=True)) must not be served the stale 16-bit twin: `lookup` checks that the parameter still lives in
    the flat buffer and `low_precision` falls back to casting the LIVE weight."""
    from segmamba_amd.param_bank import ParamBank, low_precision
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(5, 3), nn.Conv3d(4, 6, 3))
    bank = ParamBank(m, torch.bfloat16)
    w = m[1].weight
    with bank.step():
        assert bank.owns(w) and low_precision(w, torch.bfloat16).data_ptr() == bank.flat16.data_ptr() + 2 * bank.offsets[id(w)]
    m.to(memory_format=torch.channels_last_3d)                 # re-points the 5-d weight's storage (same Parameter object)
    with torch.no_grad():
        w.mul_(3.0)                                            # ... and the optimizer then updates the NEW storage
    with bank.step():
        assert not bank.owns(w)
        got = low_precision(w, torch.bfloat16)
        assert torch.equal(got, w.detach().bfloat16())          # the live values, not the bank's old copy
        assert not bank.flat16.data_ptr() <= got.data_ptr() < bank.flat16.data_ptr() + 2 * bank.flat16.numel()
        assert bank.owns(m[0].weight)                           # untouched parameters are still served from the bank
    sd = {k: v.clone() * 2 for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                          # new Parameter objects: unknown to the bank -> plain casts
    with bank.step():
        assert torch.equal(low_precision(m[0].weight, torch.bfloat16), m[0].weight.detach().bfloat16())


def test_flat_gradients_and_flat_optimizer_match_the_per_tensor_route(monkeypatch):
    """trainer flat mode (bank.attach_flat_grads + FusedClipSGD.use_flat): gradients accumulate into windows of ONE flat array
    whose addresses never change, the optimizer steps over three flat arrays; two steps equal the per-tensor route bit for bit,
    the momentum state is still per parameter, and a swapped gradient tensor sends the optimizer back to the per-tensor route."""
    from tests import emu_util
    import pytest
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    from segmamba_amd import lib as L
    monkeypatch.setattr(L, "_lib", emu_util.emu_lib())
    monkeypatch.setattr(L, "on_device", lambda t: True)
    g = torch.Generator().manual_seed(1)
    img, lab = torch.rand(1, 4, 8, 8, 8, generator=g), torch.randint(0, 4, (1, 8, 8, 8), generator=g)
    a = build_training_state(torch.device("cpu"), model=_tiny(), flat=True)
    b = build_training_state(torch.device("cpu"), model=_tiny(), flat=False)
    assert a.flat and not b.flat and a.bank.grads_attached()
    ptrs = [p.grad.data_ptr() for p in a.model.parameters()]
    for _ in range(2):
        la, lb = train_step(a, img, lab), train_step(b, img, lab)
        assert torch.equal(la, lb)
    assert [p.grad.data_ptr() for p in a.model.parameters()] == ptrs and a.optimizer.bank is a.bank
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb) and torch.equal(pa.grad, pb.grad)
        assert torch.equal(a.optimizer.state[pa]["momentum_buffer"], b.optimizer.state[pb]["momentum_buffer"])
    p0 = next(a.model.parameters())
    p0.grad = p0.grad.clone()                                   # someone replaced a gradient tensor
    a.optimizer.step()
    assert a.optimizer.bank is None                             # per-tensor route from now on, nothing silently skipped


def test_functions_take_fp32_masters_and_return_fp32_gradients():
    """linear.linear_cl / linear.pointwise Functions with fp32 weights and 16-bit activations (what autocast hands them): the
    output equals the per-parameter-cast route and the weight / bias gradients come back in fp32 (no cast-back launch), with and
    without the bank"""
    from segmamba_amd import linear as LN
    from segmamba_amd.param_bank import ParamBank
    g = torch.Generator().manual_seed(4)
    lin = nn.Linear(16, 8)
    x = torch.randn(2, 40, 16, generator=g).bfloat16()
    dy = torch.randn(2, 40, 8, generator=g).bfloat16()
    ref_w = lin.weight.detach().bfloat16().float().requires_grad_()
    ref_b = lin.bias.detach().bfloat16().float().requires_grad_()
    ref = torch.nn.functional.linear(x.float(), ref_w, ref_b)
    ref.backward(dy.float())
    bank = ParamBank(lin, torch.bfloat16)
    for use_bank in (False, True):
        lin.zero_grad()
        if use_bank:
            with bank.step():
                y = LN._LinearCL.apply(x, lin.weight, lin.bias)
                y.backward(dy)
        else:
            y = LN._LinearCL.apply(x, lin.weight, lin.bias)
            y.backward(dy)
        assert y.dtype == torch.bfloat16 and (y.float() - ref).abs().max() <= 2e-2 * float(ref.abs().max())
        assert lin.weight.grad.dtype == torch.float32 and lin.bias.grad.dtype == torch.float32
        # (a short contraction like this one is a single 16-bit GEMM; the split-K route of tall operands keeps fp32 partial sums)
        assert (lin.weight.grad - ref_w.grad).abs().max() <= 1e-2 * float(ref_w.grad.abs().max())
        assert (lin.bias.grad - ref_b.grad).abs().max() <= 1e-4 * float(ref_b.grad.abs().max())     # fp32 reduction


def test_derived_weight_packs_come_from_one_gather_and_change_nothing(monkeypatch):
    """param_bank.packed / ParamBank.derived: the packed / padded / transposed weight layouts the kernels read (3x3x3 register
    layout forward and flipped-transposed for the data gradient, padded projection weights of the Mamba block, transposed linear
    weights) are registered on first use and from the second step on are views of ONE buffer filled by ONE index_select per
    step.  Three training steps of the real network on the emulated kernels must give bit-identical parameters with the
    mechanism on and off, the packs must live in the gather buffer from step 2, and nothing new may be registered after step 1
    (a captured graph replays the step-2 launch sequence)."""
    from tests import emu_util
    import pytest
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    from segmamba_amd import conv3d as C3, lib as L, param_bank
    from segmamba_amd.mamba_simple import Mamba
    from segmamba_amd.unet_blocks import UnetResBlock
    monkeypatch.setattr(L, "_lib", emu_util.emu_lib())
    monkeypatch.setattr(L, "on_device", lambda t: True)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())      # the library's kernels wherever they apply
    from segmamba_amd import linear as LN
    monkeypatch.setattr(LN, "_ROWS_MIN", 1)                                      # ... also for 128 rows of tokens
    monkeypatch.setattr(LN, "_ROWS_HIP", True)

    class Tiny(nn.Module):                                   # one 48-channel residual block + one Mamba(v3) layer
        def __init__(self):
            super().__init__()
            self.block = UnetResBlock(48, 48)
            self.mamba = Mamba(d_model=48, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=8)

        def forward(self, v):
            y = self.block(v)                                # (1, 48, 2, 4, 16)
            t = y.flatten(2).transpose(1, 2)                 # (1, 128, 48) tokens
            return self.mamba(t)

    g = torch.Generator().manual_seed(3)
    vol = torch.randn(1, 48, 2, 4, 16, generator=g).bfloat16()     # 16-bit activations against fp32 masters: what autocast produces

    def run(enabled):
        torch.manual_seed(0)
        st = build_training_state(torch.device("cpu"), model=Tiny())
        if not enabled:
            monkeypatch.setattr(param_bank, "packed", lambda w, tag, fn: fn(w))
            monkeypatch.setattr(param_bank.ParamBank, "derived", lambda self, w, tag, fn: fn(w))
        counts = []
        for _ in range(3):
            with st.bank.step():
                st.bank.release_grads()
                st.model(vol).float().pow(2).mean().backward()
                st.bank.gather_grads()
            st.optimizer.step()
            counts.append(len(st.bank._dmaps))
        return st, counts

    on, counts = run(True)
    assert counts[0] >= 15 and counts[0] == counts[1] == counts[2], counts       # everything registered in step 1
    assert on.bank._dbuf is not None and on.bank._dready == counts[0]
    off, counts_off = run(False)
    assert counts_off == [0, 0, 0]
    for a, b in zip(on.model.parameters(), off.model.parameters()):
        assert torch.equal(a, b)


def test_fp16_flat_mode_folds_unscale_and_inf_check_into_the_fused_step(monkeypatch):
    """round 5: amp='fp16' on the flat-gradient state.  The loss scale stays a device tensor (GradScaler's own),
    the fused clip + SGD pass divides the norm by it, applies clip / scale to the scaled gradients and skips the step on inf / nan;
    the scale update is the op GradScaler.update() runs.  Same parameters as the per-tensor GradScaler route (unscale_ -> clip ->
    scaler.step -> update, reference light_training/trainer.py:461-466) - the scale is a power of two, so bit for bit."""
    from tests import emu_util
    import pytest
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    from segmamba_amd import lib as L
    monkeypatch.setattr(L, "_lib", emu_util.emu_lib())
    monkeypatch.setattr(L, "on_device", lambda t: True)
    g = torch.Generator().manual_seed(1)
    img, lab = torch.rand(1, 4, 8, 8, 8, generator=g), torch.randint(0, 4, (1, 8, 8, 8), generator=g)
    a = build_training_state(torch.device("cpu"), model=_tiny(), amp="fp16", flat=True)
    b = build_training_state(torch.device("cpu"), model=_tiny(), amp="fp16", flat=False)
    assert a.flat and a.scaler is not None and a.optimizer.loss_scale is a.scaler._scale and not b.flat
    for _ in range(2):
        la, lb = train_step(a, img, lab), train_step(b, img, lab)
        assert torch.equal(la, lb)
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb)
        assert torch.equal(a.optimizer.state[pa]["momentum_buffer"], b.optimizer.state[pb]["momentum_buffer"])
    assert float(a.found_inf) == 0.0 and a.scaler.get_scale() == b.scaler.get_scale() == 65536.0
    assert float(a.optimizer.last_clip[1]) == pytest.approx(float(torch.linalg.vector_norm(a.bank.flat_grad)) / 65536.0, rel=1e-5)
    # an overflowing step: parameters and momenta untouched, found_inf raised, scale halved - in both routes
    before = [p.detach().clone() for p in a.model.parameters()]
    mom = [a.optimizer.state[p]["momentum_buffer"].clone() for p in a.model.parameters()]
    bad = img.clone()
    bad[0, 0, 0, 0, 0] = float("inf")
    train_step(a, bad, lab), train_step(b, bad, lab)
    assert float(a.found_inf) == 1.0
    assert all(torch.equal(x, p.detach()) for x, p in zip(before, a.model.parameters()))
    assert all(torch.equal(x, a.optimizer.state[p]["momentum_buffer"]) for x, p in zip(mom, a.model.parameters()))
    assert a.scaler.get_scale() == b.scaler.get_scale() == 32768.0
    # and it goes on from there
    la, lb = train_step(a, img, lab), train_step(b, img, lab)
    assert torch.equal(la, lb) and float(a.found_inf) == 0.0
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb)
