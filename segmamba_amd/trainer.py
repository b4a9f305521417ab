"""Training-step harness for SegMamba on synthetic BraTS volumes.

Mirrors the body of the reference loop - `Trainer.train_epoch` (light_training/trainer.py:429-483) driving
`BraTSTrainer.training_step` (3_train.py:57-66):

    zero grads -> autocast forward -> CrossEntropyLoss -> backward -> clip_grad_norm_(12) -> SGD step -> poly LR

with the reference hyper-parameters (3_train.py:51-52: SGD lr 1e-2, weight decay 3e-5, momentum 0.99, nesterov;
lr_scheduler.py:36 poly 0.9).  Differences, all deliberate:
  * autocast dtype is bf16 by default (the north star's dtype).  `amp="fp16"` (or SEGM_AMP=fp16) runs the loop the reference
    actually runs - fp16 autocast with a GradScaler: scale(loss).backward(), unscale_, clip, scaler.step, scaler.update
    (light_training/trainer.py:65-67, 461-466) - through the same kernels (all instantiated for fp16);
  * data come from a device-side synthetic generator instead of the 18-process batchgenerators pipeline
    (trainer.py:154-162), which is out of scope;
  * multi-GPU is the same plain DDP (trainer.py:353-357) over RCCL; one process per GPU,
    launched by torchrun - batch per GPU stays 2 (weak scaling), gradients are all-reduced in DDP's buckets.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from .segmamba import SegMamba


class SyntheticBraTS:
    """Random 4-modality volumes + 4-class label maps generated on the device (per-rank seed as trainer.py:331)."""

    def __init__(self, batch: int, size: int, device, seed: int = 42, pool: int = 2, augment: bool = False):
        # augment: run the reference's training transforms on the device (segmamba_amd/augment.py) on every batch handed out
        self.augmenter = None
        if augment:
            from .augment import DeviceAugmenter
            self.augmenter = DeviceAugmenter(device, seed=seed)
        g = torch.Generator(device=device).manual_seed(seed)
        self.items = []
        for _ in range(pool):
            image = torch.rand(batch, 4, size, size, size, device=device, generator=g)
            if os.environ.get("SEGM_CHANNELS_LAST_3D", "0") == "1":
                image = image.contiguous(memory_format=torch.channels_last_3d)
            label = torch.randint(0, 4, (batch, size, size, size), device=device, generator=g)
            self.items.append((image, label))
        self.i = 0

    def next(self):
        item = self.items[self.i % len(self.items)]
        self.i += 1
        if self.augmenter is not None:
            item = self.augmenter(*item)
        return item


# what build_training_state passes to DistributedDataParallel (bench.py prints it in `config.ddp`).  The reference passes
# find_unused_parameters=True and the defaults otherwise (light_training/trainer.py:353-357).
DDP_SETTINGS = {"find_unused_parameters": os.environ.get("SEGM_DDP_FIND_UNUSED", "0") == "1", "gradient_as_bucket_view": True,
                "bucket_cap_mb": 64}


@dataclass
class TrainingState:
    model: nn.Module
    optimizer: torch.optim.Optimizer
    scheduler: object
    loss_fn: nn.Module
    autocast_dtype: torch.dtype = torch.bfloat16
    clip: float = 12.0
    step: int = 0
    scaler: object = None          # torch.amp.GradScaler when autocast_dtype is fp16 (reference trainer.py:65-67)
    bank: object = None            # param_bank.ParamBank: the step's 16-bit parameter copies, one launch per group


def build_training_state(device, distributed: bool = False, local_rank: int = 0, max_steps: int = 250 * 1000,
                         model: nn.Module | None = None, amp: str | None = None) -> TrainingState:
    amp = (amp or os.environ.get("SEGM_AMP", "bf16")).lower()
    if amp not in ("bf16", "fp16"):
        raise ValueError(f"amp must be 'bf16' or 'fp16', got {amp!r}")
    if model is None:
        model = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384])
    model = model.to(device)
    if os.environ.get("SEGM_CHANNELS_LAST_3D", "0") == "1":     # experiment switch: NDHWC activations / weights
        model = model.to(memory_format=torch.channels_last_3d)
    from . import lib as L
    bank = None
    if L.on_device(next(model.parameters())) and os.environ.get("SEGM_PARAM_BANK", "1") == "1":
        # the autocast copies of the weights (and the fp32 copies of their gradients) in one launch per top-level module
        # instead of one per parameter (param_bank.py); built before DDP looks at the parameters
        from .param_bank import ParamBank
        bank = ParamBank(model, torch.float16 if amp == "fp16" else torch.bfloat16)
    if distributed:
        # The reference passes find_unused_parameters=True (trainer.py:354-357); every SegMamba parameter takes part in
        # every step (DDP itself reports "did not find any unused parameters"), so the extra per-step graph traversal
        # is dropped.  Gradients live in the communication buckets (no copy in, no copy out), and the buckets are
        # large: xGMI rings are per-link bound, a few 64 MB all-reduces beat many 25 MB ones.  SEGM_DDP_FIND_UNUSED=1
        # restores the reference's flag.
        model = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[local_rank] if device.type == "cuda" else None,
            **DDP_SETTINGS)
    fused = L.on_device(next(model.parameters())) and os.environ.get("SEGM_FUSED_TRAIN_OPS", "1") != "0"
    if fused:
        # clip_grad_norm_(12) + SGD step as two passes of the library's multi-tensor kernels, cross entropy with its
        # gradient as one (csrc/trainstep.hip); SEGM_FUSED_TRAIN_OPS=0 restores the ATen calls of the reference loop
        from .train_ops import CrossEntropyLoss, FusedClipSGD
        opt = FusedClipSGD(model.parameters(), lr=1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True, max_norm=12.0)
        loss_fn = CrossEntropyLoss()
    else:
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
        loss_fn = nn.CrossEntropyLoss()
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: (1 - min(s, max_steps - 1) / max_steps) ** 0.9)
    st = TrainingState(model=model, optimizer=opt, scheduler=sched, loss_fn=loss_fn, bank=bank)
    if amp == "fp16":
        st.autocast_dtype = torch.float16
        st.scaler = torch.amp.GradScaler(device.type)     # the reference's GradScaler() defaults: 2^16, x2 / 2000 steps, x0.5 on inf
    return st


def train_step(st: TrainingState, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    if st.bank is None:
        return _train_step(st, image, label)
    with st.bank.step():                                           # forward and backward see one set of 16-bit parameter copies
        return _train_step(st, image, label)


def _train_step(st: TrainingState, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    for p in st.model.parameters():
        p.grad = None                                              # trainer.py:445
    with torch.autocast(image.device.type, dtype=st.autocast_dtype,
                        enabled=image.device.type == "cuda" or st.scaler is not None):
        pred = st.model(image)
        loss = st.loss_fn(pred, label)                             # 3_train.py:62
    fused_clip = getattr(st.optimizer, "max_norm", None) is not None    # FusedClipSGD clips inside step()
    if st.scaler is None:
        loss.backward()
        if not fused_clip:
            torch.nn.utils.clip_grad_norm_(st.model.parameters(), st.clip)  # trainer.py:464
        st.optimizer.step()
    else:                                                          # trainer.py:461-466
        st.scaler.scale(loss).backward()
        st.scaler.unscale_(st.optimizer)                           # true-scale gradients before the clip
        if not fused_clip:
            torch.nn.utils.clip_grad_norm_(st.model.parameters(), st.clip)
        st.scaler.step(st.optimizer)                               # skipped when a gradient is inf / nan
        st.scaler.update()
    st.scheduler.step()
    st.step += 1
    return loss.detach()
