"""Sliding-window inference with mirror test-time augmentation: the caller of `SegMamba.forward` at prediction time
(SURVEY.md §8f rank 2).

Host-side mirror of the reference's prediction path:
  `Predictor`                light_training/prediction.py:29-159 (`maybe_mirror_and_predict`, `predict_raw_probability`,
                             `predict_noncrop_probability`)
  `SlidingWindowInferer`     monai/inferers/inferer.py (the class 4_predict.py:55-59 and 3_train.py:35-37 construct), i.e.
  `sliding_window_inference` monai/inferers/utils.py:43-330 with its helpers `dense_patch_slices`
                             (monai/data/utils.py:171-211) and `compute_importance_map` (:1088-1138)
with the arguments the reference uses (roi 128^3, overlap 0.5, gaussian blending, sw_batch_size 1-2, 8-way mirroring).

What differs is where the data lives.  The reference moves every window batch result - and every one of the 8 mirrored
full-volume predictions - to host memory and accumulates there (prediction.py:126-152 `.cpu()` after each pass).  A
BraTS volume (4 x 240 x 240 x 155 fp32 = 143 MB, 4-class logits the same again) is nothing against 288 GB of HBM: here the
volume, the weighted accumulator, the weight map and the mirrored copies stay on the device for the whole prediction, the
eight mirror passes accumulate into one fp32 buffer, and windows go through the network in as large a batch as asked.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _tuple3(v, n: int) -> Tuple:
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def dense_patch_starts(image_size: Sequence[int], roi_size: Sequence[int], scan_interval: Sequence[int]) -> List[Tuple[int, ...]]:
    """Window origins in the reference's order (last dimension fastest; monai/data/utils.py:191-208): along every dimension
    windows start every `scan_interval`, the last one pulled back so that it ends at the image border."""
    starts = []
    for size, roi, step in zip(image_size, roi_size, scan_interval):
        if step == 0:
            num = 1
        else:
            cand = next((d for d in range(int(math.ceil(size / step))) if d * step + roi >= size), None)
            num = cand + 1 if cand is not None else 1
        starts.append([min(i * step, size - roi) for i in range(num)])
    return list(itertools.product(*starts))


def importance_map(roi_size: Sequence[int], mode: str = "constant", sigma_scale=0.125, device="cpu",
                   dtype=torch.float32) -> torch.Tensor:
    """Window weights (monai/data/utils.py:1117-1138): ones, or a separable gaussian with sigma = sigma_scale * size,
    clamped from below by max(min, 1e-3)."""
    if mode == "constant":
        w = torch.ones(tuple(roi_size), device=device, dtype=torch.float32)
    elif mode == "gaussian":
        sig = _tuple3(sigma_scale, len(roi_size))
        w = None
        for i, n in enumerate(roi_size):
            x = torch.arange(-(n - 1) / 2.0, (n - 1) / 2.0 + 1, dtype=torch.float32, device=device)
            g = torch.exp(x ** 2 / (-2 * (n * sig[i]) ** 2))
            w = g if w is None else w.unsqueeze(-1) * g[(None,) * i]
    else:
        raise ValueError(f"Unsupported mode: {mode}, available options are ['constant', 'gaussian'].")
    return torch.clamp(w, min=max(float(w.min()), 1e-3)).to(dtype)


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable[..., torch.Tensor],
                             overlap=0.25, mode: str = "constant", sigma_scale=0.125, padding_mode: str = "constant",
                             cval: float = 0.0, *args, **kwargs) -> torch.Tensor:
    """inputs (B, C, *spatial) -> predictor outputs stitched to (B, C_out, *spatial); the reference's blending:
    sum_w (weight * prediction) / sum_w weight, accumulated in the input's dtype on the input's device."""
    nd = inputs.dim() - 2
    overlap = _tuple3(overlap, nd)
    if any(o < 0 or o >= 1 for o in overlap):
        raise ValueError(f"overlap must be >= 0 and < 1, got {overlap}.")
    batch, _, *orig_size = inputs.shape
    roi = tuple(int(r) if r and r > 0 else int(s) for r, s in zip(_tuple3(roi_size, nd), orig_size))
    image_size = tuple(max(s, r) for s, r in zip(orig_size, roi))
    pad = []
    for k in range(nd - 1, -1, -1):                        # images smaller than the window are centred in padding
        diff = max(roi[k] - orig_size[k], 0)
        pad.extend([diff // 2, diff - diff // 2])
    if any(pad):
        inputs = F.pad(inputs, pad, mode=padding_mode, value=cval)
    interval = tuple(r if r == s else max(int(r * (1 - o)), 1) for r, s, o in zip(roi, image_size, overlap))
    starts = dense_patch_starts(image_size, roi, interval)
    weight = importance_map(roi, mode, sigma_scale, inputs.device, inputs.dtype)[None, None]

    out: Optional[torch.Tensor] = None
    count = torch.zeros((1, 1) + image_size, dtype=inputs.dtype, device=inputs.device)
    for st in starts:
        count[(slice(None), slice(None)) + tuple(slice(s, s + r) for s, r in zip(st, roi))] += weight
    jobs = [(b, st) for b in range(batch) for st in starts]
    for j0 in range(0, len(jobs), sw_batch_size):
        chunk = jobs[j0:j0 + sw_batch_size]
        win = torch.cat([inputs[(slice(b, b + 1), slice(None)) + tuple(slice(s, s + r) for s, r in zip(st, roi))]
                         for b, st in chunk])
        pred = predictor(win, *args, **kwargs)
        if pred.shape[2:] != roi:
            raise RuntimeError("sliding_window_inference: the predictor must keep the window's spatial size")
        if out is None:
            out = torch.zeros((batch, pred.shape[1]) + image_size, dtype=inputs.dtype, device=inputs.device)
        pred = pred.to(inputs.dtype) * weight
        for i, (b, st) in enumerate(chunk):
            out[(slice(b, b + 1), slice(None)) + tuple(slice(s, s + r) for s, r in zip(st, roi))] += pred[i:i + 1]
    out = out / count
    if any(pad):                                            # remove the centring padding again
        sl = [slice(None), slice(None)]
        for d in range(nd):
            p0 = pad[2 * (nd - 1 - d)]
            sl.append(slice(p0, p0 + orig_size[d]))
        out = out[tuple(sl)]
    return out


class SlidingWindowInferer:
    """Constructor-compatible with the object the reference builds (4_predict.py:55-59, 3_train.py:35-37)."""

    def __init__(self, roi_size, sw_batch_size: int = 1, overlap=0.25, mode: str = "constant", sigma_scale=0.125,
                 padding_mode: str = "constant", cval: float = 0.0, progress: bool = False, **unused) -> None:
        self.roi_size, self.sw_batch_size, self.overlap = roi_size, sw_batch_size, overlap
        self.mode, self.sigma_scale, self.padding_mode, self.cval = mode, sigma_scale, padding_mode, cval

    def __call__(self, inputs: torch.Tensor, network: Callable[..., torch.Tensor], *args, **kwargs) -> torch.Tensor:
        return sliding_window_inference(inputs, self.roi_size, self.sw_batch_size, network, self.overlap, self.mode,
                                        self.sigma_scale, self.padding_mode, self.cval, *args, **kwargs)


class Predictor:
    """reference light_training/prediction.py:29-159, device-resident."""

    def __init__(self, window_infer, mirror_axes=None, autocast_dtype: torch.dtype = torch.bfloat16) -> None:
        """`autocast_dtype`: the reference runs the network under `torch.autocast("cuda")`, i.e. fp16 (prediction.py:124);
        on MI355X the path's 16-bit type is bf16 (BASELINE north star), so that is the default.  Pass torch.float16 for
        the reference's behaviour (same kernels, same speed)."""
        self.window_infer = window_infer
        self.mirror_axes = mirror_axes
        self.autocast_dtype = autocast_dtype

    def maybe_mirror_and_predict(self, x: torch.Tensor, model, device=torch.device("cpu"), **kwargs) -> torch.Tensor:
        """Mean over the 2^len(mirror_axes) mirrored sliding-window predictions (reference :110-159).  The sum runs in the
        reference's order (identity, each single axis, each pair, all three) in fp32 on `device`; the result stays there."""
        device = torch.device(device) if isinstance(device, str) else device
        model.to(device)
        x = x.to(device)
        axes = self.mirror_axes
        if axes is not None and len(axes) and max(axes) > x.dim() - 3:
            raise AssertionError("mirror_axes does not match the dimension of the input!")
        combos: List[Tuple[int, ...]] = [()]
        if axes is not None:
            ordered = sorted(axes)
            for k in range(1, len(ordered) + 1):
                combos += list(itertools.combinations(ordered, k))
        with torch.no_grad(), torch.autocast("cuda", dtype=self.autocast_dtype,
                                             enabled=device.type == "cuda" and self.autocast_dtype != torch.float32):
            total = None
            for c in combos:
                dims = tuple(a + 2 for a in c)
                xin = torch.flip(x, dims) if dims else x
                p = self.window_infer(xin, model, **kwargs).float()
                p = torch.flip(p, dims) if dims else p
                total = p if total is None else total + p
            return total / len(combos)

    @staticmethod
    def predict_raw_probability(model_output: torch.Tensor, properties: dict) -> torch.Tensor:
        """trilinear resampling back to the pre-resample crop shape, channel by channel (reference :33-62)."""
        if model_output.dim() == 5:
            model_output = model_output[0]
        d, w, h = (int(v) for v in properties["shape_after_cropping_before_resample"][:3])
        with torch.no_grad():
            return torch.stack([F.interpolate(model_output[c][None, None].float(), size=(d, w, h), mode="trilinear")[0, 0]
                                for c in range(model_output.shape[0])]).to(model_output.dtype)

    @staticmethod
    def predict_noncrop_probability(model_output, properties: dict) -> np.ndarray:
        """paste the cropped prediction back into the full-size uint8 volume (reference :64-108)."""
        if isinstance(model_output, torch.Tensor):
            model_output = model_output.cpu().numpy()
        shape = [int(v.item()) if isinstance(v, torch.Tensor) else int(v) for v in properties["shape_before_cropping"][:3]]
        (a0, a1), (b0, b1), (c0, c1) = [tuple(int(v) for v in bb) for bb in properties["bbox_used_for_cropping"][:3]]
        if model_output.ndim == 3:
            full = np.zeros(shape, dtype=np.uint8)
            full[a0:a1, b0:b1, c0:c1] = model_output
        elif model_output.ndim == 4:
            full = np.zeros([model_output.shape[0]] + shape, dtype=np.uint8)
            full[:, a0:a1, b0:b1, c0:c1] = model_output
        else:
            raise ValueError("restore crop error: expected a 3-D label map or a 4-D (C, ...) volume")
        return full
