"""Device-side training augmentation: the feeder that replaces the reference's 18-process batchgenerators pipeline.

The reference augments on the host (light_training/trainer.py:154-162 starts NonDetMultiThreadedAugmenter workers running
`get_train_transforms`, light_training/augment/train_augment.py:24-66) and ships each batch over PCIe.  Here the batch is
already resident on the GPU (`SyntheticBraTS`, or a real loader's pinned upload) and the same chain runs there as a
handful of elementwise / resampling ops per step - a few hundred microseconds, no worker processes, no host round trip.

The chain, in the reference's order and with its parameters (train_augment.py:30-60).  `batchgenerators` itself is a
third-party dependency that is NOT in the reference tree (setup: pip `batchgenerators`, unpinned); what each transform does
is restated from its published implementation:

  SpatialTransform         rotation about x / y / z by U(-30 deg, 30 deg) with p 0.2, isotropic scale with p 0.2 drawn two-sided - U(0.7, 1) or U(1, 1.4)
                           with equal odds, batchgenerators' rule for a range straddling 1 -
                           about the patch centre, zero padding; image cubic spline / label linear in the reference ->
                           trilinear / nearest here (grid_sample has no 3-D cubic mode); labels outside the volume become 0
                           (RemoveLabelTransform(-1, 0), :57)
  GaussianNoise            p 0.1: x += N(0, s), s ~ U(0, 0.1)   (batchgenerators passes its "variance" as numpy's scale)
  GaussianBlur             p 0.2, each channel with p 0.5: separable gaussian, sigma ~ U(0.5, 1) per channel; the kernel is cut at
                           radius 3 (>= 3 sigma) and the border replicated, where scipy's gaussian_filter cuts at 4 sigma and
                           reflects: differences below 1e-3 of the kernel mass, at the border voxels only
  BrightnessMultiplicative p 0.15: each channel x U(0.75, 1.25)
  ContrastAugmentation     p 0.15: each channel (x - mean) f + mean, f from (0.75, 1) or (1, 1.25) with equal odds, clipped
                           to the channel's former range
  SimulateLowResolution    p 0.25, each channel with p 0.5: nearest down by U(0.5, 1), back up (cubic in the reference ->
                           trilinear here)
  Gamma (inverted image)   p 0.1;  Gamma  p 0.3: per channel, gamma from (0.7, 1) or (1, 1.5) with equal odds on the
                           range-normalised channel, mean / std retained
  Mirror                   each of the three axes with p 0.5, image and label together

Every random decision comes from one torch.Generator on the device (reproducible per rank: seed 42 + rank, trainer.py:331).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class DeviceAugmenter:
    def __init__(self, device, seed: int = 42, mirror_axes=(0, 1, 2), spatial: bool = True):
        self.g = torch.Generator(device=device).manual_seed(seed)
        self.device = torch.device(device)
        self.mirror_axes = tuple(mirror_axes or ())
        self.spatial = spatial

    # ---- random helpers (all on the device generator) ------------------------------------------------------------------
    def _u(self, lo, hi, *shape):
        return lo + (hi - lo) * torch.rand(*shape, device=self.device, generator=self.g)

    def _coin(self, p, *shape):
        return torch.rand(*shape, device=self.device, generator=self.g) < p

    def _two_sided(self, lo, hi, *shape):
        """batchgenerators' contrast / gamma sampling: (lo, 1) or (max(lo, 1), hi) with equal odds"""
        low = self._u(lo, 1.0, *shape)
        high = self._u(max(lo, 1.0), hi, *shape)
        return torch.where(self._coin(0.5, *shape), low, high)

    # ---- transforms -------------------------------------------------------------------------------------------------------
    def _spatial(self, x, y):
        B = x.shape[0]
        rot = self._coin(0.2, B)
        scl = self._coin(0.2, B)
        if not bool((rot | scl).any()):
            return x, y
        a = self._u(-math.pi / 6, math.pi / 6, B, 3) * rot[:, None]
        s = torch.where(scl, self._two_sided(0.7, 1.4, B), torch.ones(B, device=self.device))
        cx, sx, cy, sy, cz, sz = a[:, 0].cos(), a[:, 0].sin(), a[:, 1].cos(), a[:, 1].sin(), a[:, 2].cos(), a[:, 2].sin()
        one, zero = torch.ones_like(cx), torch.zeros_like(cx)
        Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).view(B, 3, 3)
        Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).view(B, 3, 3)
        Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).view(B, 3, 3)
        M = (Rz @ Ry @ Rx) * s[:, None, None]              # output coordinate -> input coordinate (scale > 1 zooms out)
        theta = torch.cat([M, torch.zeros(B, 3, 1, device=self.device)], 2)
        grid = F.affine_grid(theta, list(x.shape), align_corners=False)
        x = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        y = F.grid_sample(y[:, None].float(), grid, mode="nearest", padding_mode="zeros", align_corners=False)[:, 0].to(y.dtype)
        return x, y

    def _blur(self, x):
        B, C = x.shape[:2]
        on = self._coin(0.2, B)[:, None] & self._coin(0.5, B, C)
        if not bool(on.any()):
            return x
        sigma = self._u(0.5, 1.0, B, C)
        r = 3                                               # 3 sigma at sigma <= 1
        t = torch.arange(-r, r + 1, device=self.device, dtype=x.dtype)
        k = torch.exp(-0.5 * (t[None, None] / sigma[:, :, None]) ** 2)
        k = k / k.sum(-1, keepdim=True)                      # (B, C, 7)
        ident = torch.zeros_like(k)
        ident[:, :, r] = 1
        k = torch.where(on[:, :, None], k, ident).reshape(B * C, 1, 2 * r + 1)
        v = x.reshape(1, B * C, *x.shape[2:])
        for dim in range(3):                                # separable: one 1-D depthwise convolution per axis
            shape = [B * C, 1, 1, 1, 1]
            shape[2 + dim] = 2 * r + 1
            pad = [0, 0, 0, 0, 0, 0]
            pad[2 * (2 - dim)] = pad[2 * (2 - dim) + 1] = r
            v = F.conv3d(F.pad(v, pad, mode="replicate"), k.reshape(shape), groups=B * C)
        return v.reshape(x.shape)

    def _low_res(self, x):
        B, C = x.shape[:2]
        on = self._coin(0.25, B)[:, None] & self._coin(0.5, B, C)
        if not bool(on.any()):
            return x
        zoom = self._u(0.5, 1.0, B, C)
        out = x.clone()
        for b, c in on.nonzero().tolist():                  # few (sample, channel) pairs per step; each has its own grid size
            size = [max(1, int(round(d * float(zoom[b, c])))) for d in x.shape[2:]]
            small = F.interpolate(x[b:b + 1, c:c + 1], size=size, mode="nearest")
            out[b, c] = F.interpolate(small, size=list(x.shape[2:]), mode="trilinear", align_corners=False)[0, 0]
        return out

    def _gamma(self, x, p, invert):
        B, C = x.shape[:2]
        on = self._coin(p, B)
        if not bool(on.any()):
            return x
        v = -x if invert else x
        red = (2, 3, 4)
        mn, sd = v.mean(red, keepdim=True), v.std(red, keepdim=True)
        lo = v.amin(red, keepdim=True)
        rng = v.amax(red, keepdim=True) - lo
        gam = self._two_sided(0.7, 1.5, B, C)[:, :, None, None, None]
        w = ((v - lo) / (rng + 1e-7)).clamp_min(0).pow(gam) * rng + lo
        w = w - w.mean(red, keepdim=True)
        w = w / (w.std(red, keepdim=True) + 1e-8) * sd + mn
        if invert:
            w = -w
        return torch.where(on[:, None, None, None, None], w, x)

    def __call__(self, image: torch.Tensor, label: torch.Tensor):
        """image (B, C, D, H, W) float, label (B, D, H, W) integer class map -> augmented copies (same shapes / dtypes)"""
        x, y = image, label
        B, C = x.shape[:2]
        if self.spatial:
            x, y = self._spatial(x, y)
        noise = self._coin(0.1, B)
        if bool(noise.any()):
            s = self._u(0.0, 0.1, B) * noise
            x = x + torch.randn(x.shape, device=self.device, generator=self.g, dtype=x.dtype) * s[:, None, None, None, None]
        x = self._blur(x)
        bright = self._coin(0.15, B)
        x = x * torch.where(bright[:, None], self._u(0.75, 1.25, B, C), torch.ones(B, C, device=self.device))[:, :, None, None, None]
        con = self._coin(0.15, B)
        if bool(con.any()):
            red = (2, 3, 4)
            mn, lo, hi = x.mean(red, keepdim=True), x.amin(red, keepdim=True), x.amax(red, keepdim=True)
            f = self._two_sided(0.75, 1.25, B, C)[:, :, None, None, None]
            x = torch.where(con[:, None, None, None, None], torch.minimum(torch.maximum((x - mn) * f + mn, lo), hi), x)
        x = self._low_res(x)
        x = self._gamma(x, 0.1, invert=True)
        x = self._gamma(x, 0.3, invert=False)
        for ax in self.mirror_axes:
            flip = self._coin(0.5, B)
            if bool(flip.any()):
                x = torch.where(flip[:, None, None, None, None], x.flip(2 + ax), x)
                y = torch.where(flip[:, None, None, None], y.flip(1 + ax), y)
        return x.contiguous(), y.contiguous()
