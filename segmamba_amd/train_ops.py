"""Device-side training-step glue: fused cross entropy and clip + SGD, on the HIP library.

  * `cross_entropy(logits, labels)` == `nn.CrossEntropyLoss()(logits, labels)` (reference 3_train.py:48,62; mean
    reduction, ignore_index -100) as one kernel that also produces d(loss)/d(logits);
  * `FusedClipSGD` == `torch.nn.utils.clip_grad_norm_(params, max_norm)` + `torch.optim.SGD(..., nesterov=True).step()`
    (reference light_training/trainer.py:461-470, 3_train.py:51-52) in two passes over the parameters.  State layout
    (`momentum_buffer` per parameter) and `param_groups` are those of torch.optim.SGD, so checkpoints interchange.

CPU tensors (the unit tests of the host logic) take the ATen path; CUDA tensors always call the library.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn.functional as F


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        from . import lib as L, ops_raw
        loss_sum, count, dlogits = ops_raw.cross_entropy(L.get_lib(), logits, labels, ignore_index)
        ctx.save_for_backward(dlogits, count)
        return loss_sum / count

    @staticmethod
    def backward(ctx, grad):
        dlogits, count = ctx.saved_tensors
        return dlogits * (grad / count).to(dlogits.dtype), None, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Mean cross entropy over the class axis (dim 1); fp32 scalar.  Under autocast the logits are used in the dtype they
    arrive in (the kernel computes in fp32 either way), which skips ATen's fp32 copy of the logits."""
    from . import lib as L
    if not L.on_device(logits):
        return F.cross_entropy(logits.float(), labels, ignore_index=ignore_index)
    from . import ops_raw
    if not ops_raw.cross_entropy_supported(logits, labels):
        raise RuntimeError("cross_entropy: expected logits (B, C <= 16, *spatial) fp32 / fp16 / bf16 and int64 labels (B, *spatial)")
    with torch.autocast("cuda", enabled=False):
        return _CrossEntropy.apply(logits, labels, ignore_index)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the `nn.CrossEntropyLoss()` of the reference trainer (3_train.py:48)."""

    def __init__(self, ignore_index: int = -100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, labels):
        return cross_entropy(logits, labels, self.ignore_index)


class FusedClipSGD(torch.optim.Optimizer):
    """SGD(momentum, nesterov, weight_decay, dampening 0) whose `step()` also applies `clip_grad_norm_(max_norm)` over ALL
    its parameters first (max_norm None / <= 0: no clipping).  The gradients themselves are left unscaled."""

    def __init__(self, params: Iterable, lr: float = 1e-2, momentum: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = False, max_norm: Optional[float] = None):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov,
                                      dampening=0, maximize=False, foreach=None, differentiable=False, fused=None))
        self.max_norm = max_norm
        self.last_clip = None                 # device tensor {clip coefficient, gradient norm, -, -} of the last step
        self.bank = None                      # param_bank.ParamBank with flat gradients attached: see use_flat()
        self._flat_mom = None
        # fp16 autocast (the reference's GradScaler loop, light_training/trainer.py:461-466) without passes over the gradients:
        # `loss_scale` = the 1-element fp32 tensor holding the factor the gradients carry (GradScaler's own scale tensor); step()
        # then clips by the norm of the UNSCALED gradients, updates with clip / scale, and skips the whole step - parameters and
        # momenta untouched, 1 written to `found_inf` - when a gradient is inf / nan (what unscale_ + scaler.step do)
        self.loss_scale = None
        self.found_inf = None

    def use_loss_scale(self, scale: torch.Tensor) -> torch.Tensor:
        """-> found_inf (1-element fp32, rewritten by every step()): the tensor torch._amp_update_scale_ takes"""
        if scale.dtype != torch.float32 or scale.numel() != 1:
            raise RuntimeError("FusedClipSGD.use_loss_scale: a 1-element fp32 tensor (GradScaler's scale) is required")
        self.loss_scale = scale
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=scale.device)
        return self.found_inf

    def use_flat(self, bank) -> None:
        """Step over the bank's three flat arrays (parameters, gradients, momenta) - ONE tensor per launch instead of a table of
        286 pointers, and nothing for the host to gather per step.  Needs `bank.attach_flat_grads()` and one parameter group
        holding exactly the bank's parameters; `state[p]["momentum_buffer"]` stays a per-parameter tensor (a window of the flat
        momentum array), so the state dict still interchanges with torch.optim.SGD."""
        ps = [p for gr in self.param_groups for p in gr["params"]]
        if len(self.param_groups) != 1 or len(ps) != len(bank.params) or {id(p) for p in ps} != {id(p) for p in bank.params}:
            raise RuntimeError("FusedClipSGD.use_flat: one parameter group holding exactly the bank's parameters is required")
        if bank.flat_grad is None:
            raise RuntimeError("FusedClipSGD.use_flat: call bank.attach_flat_grads() first")
        self._flat_mom = torch.zeros_like(bank.flat32)
        for p in bank.params:
            o = bank.offsets[id(p)]
            win = self._flat_mom[o:o + p.numel()].view(p.shape)
            old = self.state[p].get("momentum_buffer")
            if old is not None:
                win.copy_(old)
            self.state[p]["momentum_buffer"] = win
        self.bank = bank

    def _flat_ok(self) -> bool:
        """the flat arrays still are what the parameters, their gradients and their momenta live in"""
        b = self.bank
        if b is None or not b.grads_attached():
            return False
        mbase = self._flat_mom.data_ptr()
        for p in b.params:
            m = self.state[p].get("momentum_buffer")
            if m is None or m.data_ptr() != mbase + 4 * b.offsets[id(p)] or not b.owns(p):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.bank is not None:
            if not self._flat_ok():               # someone re-pointed a parameter / gradient / momentum: per-tensor route
                self.bank = None
            else:
                from . import lib as L, ops_raw
                gr = self.param_groups[0]
                self.last_clip = ops_raw.sgd_clip_step(L.get_lib(), [self.bank.flat32], [self.bank.flat_grad], [self._flat_mom],
                                                       gr["lr"], gr["momentum"], gr["weight_decay"], gr["nesterov"],
                                                       float(self.max_norm) if self.max_norm else 0.0,
                                                       loss_scale=self.loss_scale, found_inf=self.found_inf)
                return loss
        groups = []
        for gr in self.param_groups:
            ps = [p for p in gr["params"] if p.grad is not None]
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            groups.append((gr, ps))
        allp = [p for _, ps in groups for p in ps]
        if not allp:
            return loss
        max_norm = float(self.max_norm) if self.max_norm else 0.0
        from . import lib as L
        if L.on_device(allp[0]):
            self._step_hip(groups, allp, max_norm)
        else:
            self._step_aten(groups, allp, max_norm)
        return loss

    def _step_hip(self, groups, allp, max_norm):
        from . import lib as L, ops_raw
        hip = L.get_lib()
        same = len({(g["lr"], g["momentum"], g["weight_decay"], g["nesterov"]) for g, _ in groups}) == 1
        if not same and (max_norm > 0 or self.loss_scale is not None):
            raise RuntimeError("FusedClipSGD: clipping / loss scaling need one set of hyper-parameters over all groups")
        for gr, ps in ([(groups[0][0], allp)] if same else groups):
            if not ps:
                continue
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            self.last_clip = ops_raw.sgd_clip_step(hip, [p.data for p in ps], grads, [self.state[p]["momentum_buffer"] for p in ps],
                                                   gr["lr"], gr["momentum"], gr["weight_decay"], gr["nesterov"], max_norm,
                                                   loss_scale=self.loss_scale, found_inf=self.found_inf)

    def _step_aten(self, groups, allp, max_norm):
        coef = 1.0
        inv = 1.0 if self.loss_scale is None else 1.0 / self.loss_scale
        if max_norm > 0 or self.loss_scale is not None:
            raw = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in allp]))
            if self.loss_scale is not None:
                bad = not bool(torch.isfinite(raw))
                self.found_inf.fill_(1.0 if bad else 0.0)
                if bad:
                    return
            if max_norm > 0:
                coef = torch.clamp(max_norm / (raw * inv + 1e-6), max=1.0)
        coef = coef * inv
        for gr, ps in groups:
            for p in ps:
                g = p.grad * coef + gr["weight_decay"] * p
                m = self.state[p]["momentum_buffer"]
                m.mul_(gr["momentum"]).add_(g)
                p.sub_(gr["lr"] * (g + gr["momentum"] * m if gr["nesterov"] else m))
