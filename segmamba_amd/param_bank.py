"""Low-precision copies of the parameters, made once per step by one launch instead of one per parameter.

Under autocast the reference casts every fp32 weight to the compute dtype where it is used and casts its gradient back
(`at::autocast::cached_cast` / `ToCopyBackward`, one element-wise kernel each): for SegMamba ~280 + ~210 launches of a few
microseconds per training step - 2.1 ms of a 70 ms step on MI355X (profiles/r02_step_kernels_final.txt: 280
`bfloat16_copy_kernel` + 213 `bfloat16tofloat32_copy_kernel` launches).  Two changes remove them:

  * the host modules' autograd Functions (linear.py, conv3d.py, fused_norm.py, selective_scan_interface.py) take the fp32
    parameter itself, make the 16-bit copy INSIDE their forward (`low_precision`, no autograd node) and return the weight
    gradient in fp32 - their split-K / MFMA weight-gradient kernels accumulate in fp32 anyway - so nothing is cast back
    (and the gradient is not rounded to 16 bits on the way, as the reference's is);
  * `ParamBank(model, dtype)` moves the fp32 parameters into one flat buffer (`p.data` becomes a view of it: `state_dict`,
    the optimizer and DDP see the same tensors) next to a 16-bit buffer of the same layout; `with bank.step():` converts the
    whole buffer with ONE copy kernel and `low_precision(p, dtype)` hands out views of it - also for views of a parameter
    (a reshaped / transposed / channel-sliced weight), which are the same `as_strided` window of the 16-bit buffer.

Outside `bank.step()` (inference, the reference's own loop without the harness) `low_precision` is `p.to(dtype)`: nothing
depends on the bank being there.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional

import torch
import torch.nn as nn

_ACTIVE: Optional["ParamBank"] = None


class ParamBank:
    def __init__(self, model: nn.Module, dtype: torch.dtype):
        self.dtype = dtype
        params, seen = [], set()
        for p in model.parameters():
            if p.dtype == torch.float32 and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("ParamBank: the model has no fp32 parameters")
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8                  # 16-byte aligned 16-bit views (MFMA operand loads)
        self.flat32 = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat16 = torch.zeros(total, dtype=dtype, device=dev)
        self.views16: Dict[int, torch.Tensor] = {}
        self.offsets: Dict[int, int] = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = self.flat32[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.views16[id(p)] = self.flat16[o:o + p.numel()].view(p.shape)
                self.offsets[id(p)] = o
        self.params = params                                   # keeps the ids alive
        self.fresh = False
        self.flat_grad: Optional[torch.Tensor] = None

    def attach_flat_grads(self) -> torch.Tensor:
        """`p.grad` of every parameter becomes a window of ONE flat fp32 buffer laid out like `flat32` (what DDP's
        `gradient_as_bucket_view` does per bucket): autograd accumulates in place, so the addresses never change - a captured
        step (trainer.GraphedStep) replays into them, a data-parallel run all-reduces the single buffer, and the optimizer
        walks three flat arrays instead of 286 tensors.  Zero it (`flat_grad.zero_()`) where the loop set `p.grad = None`."""
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat32)
        for p in self.params:
            o = self.offsets[id(p)]
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        return self.flat_grad

    def grads_attached(self) -> bool:
        """every parameter's gradient still is its window of `flat_grad` (nobody set `p.grad = None` or swapped it)"""
        if self.flat_grad is None:
            return False
        base = self.flat_grad.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == base + 4 * self.offsets[id(p)] for p in self.params)

    def owns(self, p: torch.Tensor) -> bool:
        """`p.data` still is the window of `flat32` the bank gave it.  Anything that re-points the storage after the bank was
        built - `model.to(memory_format=...)`, `.to(dtype / device)`, `load_state_dict(assign=True)` - makes the 16-bit twin
        stale for that parameter; `lookup` then declines and the caller casts the live weight instead."""
        o = self.offsets.get(id(p))
        return (o is not None and p.dtype == torch.float32 and p.device == self.flat32.device and
                p.data_ptr() == self.flat32.data_ptr() + 4 * o and p.is_contiguous())

    def refresh(self) -> None:
        with torch.no_grad():
            self.flat16.copy_(self.flat32)                     # every parameter, one launch
        self.fresh = True

    def lookup(self, w: torch.Tensor) -> Optional[torch.Tensor]:
        base = w._base if w._is_view() else w
        v = self.views16.get(id(base))
        if v is None or not self.fresh or not self.owns(base):
            return None
        if w is base:
            return v
        if base.dtype != torch.float32 or w.dtype != torch.float32:
            return None
        # as_strided's offset counts from the start of the STORAGE: the parameter's place in the flat buffer + the view's own
        return v.as_strided(w.size(), w.stride(), v.storage_offset() + w.storage_offset() - base.storage_offset())

    @contextlib.contextmanager
    def step(self):
        """the forward AND backward pass of one training step: the 16-bit copies are made on entry; they go stale when the
        optimizer moves the parameters, so they are not handed out after the block"""
        global _ACTIVE
        prev, _ACTIVE = _ACTIVE, self
        self.refresh()
        try:
            yield self
        finally:
            _ACTIVE = prev
            self.fresh = False


def low_precision(p: Optional[torch.Tensor], dtype: torch.dtype) -> Optional[torch.Tensor]:
    """`p` in `dtype` WITHOUT an autograd edge (callers are autograd Functions that return the gradient of `p` themselves): the
    active bank's copy for a parameter (or a view of one) it holds, else a plain cast."""
    if p is None:
        return None
    if p.dtype == dtype:
        return p.detach()
    if _ACTIVE is not None and dtype == _ACTIVE.dtype:
        v = _ACTIVE.lookup(p)
        if v is not None:
            return v
    return p.detach().to(dtype)
