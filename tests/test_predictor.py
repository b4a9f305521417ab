"""The prediction-time caller of the hot path (sliding window + mirror TTA) against fixtures generated
by the reference's own MONAI SlidingWindowInferer + Predictor (tests/golden/make_golden_predict.py)."""
import os

import numpy as np
import pytest
import torch

from segmamba_amd import predictor as P
from tests.golden.make_golden_predict import CASES, toy_net

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "predict.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_sliding_window_and_mirror_tta_match_reference(name):
    shape, roi, swb, ov, mode, axes = CASES[name]
    x = torch.rand(shape, generator=torch.Generator().manual_seed(sum(shape)))
    net = toy_net(shape[1], 3)
    inferer = P.SlidingWindowInferer(roi_size=roi, sw_batch_size=swb, overlap=ov, mode=mode, progress=True)
    with torch.no_grad():
        win = inferer(x, net)
    assert np.allclose(win.numpy(), GOLD[name + "_window"], atol=2e-6)
    tta = P.Predictor(window_infer=inferer, mirror_axes=axes).maybe_mirror_and_predict(x, net)
    assert np.allclose(tta.numpy(), GOLD[name + "_tta"], atol=2e-6)
    # a different window batch size changes nothing
    other = P.SlidingWindowInferer(roi_size=roi, sw_batch_size=5, overlap=ov, mode=mode)
    with torch.no_grad():
        assert torch.allclose(other(x, net), win, atol=1e-6)


def test_window_origins_and_weights():
    # reference: windows every int(roi * (1 - overlap)), the last one flush with the border; last dimension fastest
    assert P.dense_patch_starts((20, 16), (16, 16), (8, 16)) == [(0, 0), (4, 0)]
    assert P.dense_patch_starts((33,), (16,), (8,)) == [(0,), (8,), (16,), (17,)]
    w = P.importance_map((16, 16, 16), "gaussian")
    assert w.shape == (16, 16, 16) and float(w.max()) <= 1.0 and float(w.min()) >= 1e-3
    assert torch.equal(P.importance_map((4, 4), "constant"), torch.ones(4, 4))
    with pytest.raises(ValueError):
        P.importance_map((4,), "cosine")
    with pytest.raises(ValueError):
        P.sliding_window_inference(torch.zeros(1, 1, 8, 8, 8), 4, 1, lambda t: t, overlap=1.0)


def test_resample_and_uncrop_match_reference():
    props = {"shape_after_cropping_before_resample": [11, 9, 14], "shape_before_cropping": [20, 18, 22],
             "bbox_used_for_cropping": [[3, 14], [5, 14], [2, 16]]}
    logits = torch.rand(1, 3, 8, 8, 8, generator=torch.Generator().manual_seed(3))
    raw = P.Predictor.predict_raw_probability(logits, props)
    assert np.allclose(raw.numpy(), GOLD["raw_probability"], atol=1e-3)          # the reference stores fp16
    full = P.Predictor.predict_noncrop_probability(torch.from_numpy(GOLD["raw_probability"]).argmax(0).to(torch.uint8), props)
    assert full.dtype == np.uint8 and np.array_equal(full, GOLD["noncrop"])
    with pytest.raises(ValueError):
        P.Predictor.predict_noncrop_probability(np.zeros((2, 2)), props)


@pytest.mark.gpu
def test_predictor_on_device_full_brats_case():
    """A BraTS-sized case (4 x 155 x 240 x 240 cropped to 4 x 138 x 176 x 144) through SegMamba with 128^3 windows,
    overlap 0.5, gaussian blending and 8-way mirroring, everything resident on the GPU: finite, on the device, and equal
    to the un-mirrored prediction of the mirrored input (flip consistency of the whole pipeline)."""
    from segmamba_amd.segmamba import SegMamba
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384]).to(dev).eval()
    x = torch.rand(1, 4, 138, 176, 144, device=dev)
    inferer = P.SlidingWindowInferer(roi_size=[128, 128, 128], sw_batch_size=2, overlap=0.5, mode="gaussian")
    pred = P.Predictor(window_infer=inferer, mirror_axes=[0, 1, 2])
    y = pred.maybe_mirror_and_predict(x, model, device=dev)
    assert y.shape == (1, 4, 138, 176, 144) and y.device.type == "cuda" and torch.isfinite(y).all()
    y_flip = pred.maybe_mirror_and_predict(torch.flip(x, (3,)), model, device=dev)
    assert (torch.flip(y_flip, (3,)) - y).abs().max() <= 2e-2 * float(y.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_sliding_window_and_mirror_tta_match_reference_on_the_gpu(name):
    """the reference-generated fixtures again, with the volume, the window batches, the blending accumulator and the mirrored
    predictions resident on the GPU (the setting the predictor is written for)"""
    shape, roi, swb, ov, mode, axes = CASES[name]
    x = torch.rand(shape, generator=torch.Generator().manual_seed(sum(shape))).cuda()
    net = toy_net(shape[1], 3).cuda()
    inferer = P.SlidingWindowInferer(roi_size=roi, sw_batch_size=swb, overlap=ov, mode=mode)
    with torch.no_grad():
        win = inferer(x, net)
    assert win.is_cuda
    # the toy network's convolutions run on MIOpen here and on ATen's CPU kernels in the fixture: fp32 round-off of a different
    # summation order (per mirrored input a different solver may be picked), bounded relative to the output's magnitude
    def err(got, key):
        ref = GOLD[key]
        return float(np.abs(got.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
    e_win = err(win, name + "_window")
    assert e_win <= 2e-5, f"window prediction: relative error {e_win:.3e}"
    tta = P.Predictor(window_infer=inferer, mirror_axes=axes).maybe_mirror_and_predict(x, net, device=torch.device("cuda"))
    e_tta = err(tta, name + "_tta")
    # on the GPU the Predictor runs the network under 16-bit autocast, as the reference does (prediction.py:124): bf16 rounding
    assert tta.is_cuda and e_tta <= 2e-2, f"mirror TTA: relative error {e_tta:.3e}"
    fp32 = P.Predictor(window_infer=inferer, mirror_axes=axes, autocast_dtype=torch.float32)
    e32 = err(fp32.maybe_mirror_and_predict(x, net, device=torch.device("cuda")), name + "_tta")
    assert e32 <= 2e-5, f"mirror TTA without autocast: relative error {e32:.3e}"
