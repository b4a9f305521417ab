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

from . import lib as _lib

_ACTIVE: Optional["ParamBank"] = None


def _ops_raw():
    from . import ops_raw
    return ops_raw


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
        # the 16-bit twin sits behind 8 zero elements (16 bytes: the views stay 16-byte aligned): index 0 of `flat16z` is the
        # "zero" every padding element of a derived pack gathers (see `packed`)
        self.flat16z = torch.zeros(total + 8, dtype=dtype, device=dev)
        self.flat16 = self.flat16z[8:]
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
        # derived packs (see `packed`): re-arranged copies of weights, all produced by ONE gather per step
        self._iota: Optional[torch.Tensor] = None              # int32: position of every flat16 element inside flat16z
        self._dkeys: Dict[tuple, tuple] = {}                   # key -> (offset, shape) inside _dbuf
        self._dmaps: list = []                                 # per pack: int32 source indices (0 = the zero element)
        self._dtotal = 0
        self._dmap: Optional[torch.Tensor] = None              # the maps concatenated (rebuilt when a pack was registered)
        self._dbuf: Optional[torch.Tensor] = None
        self._dcompact: Optional[torch.Tensor] = None          # the maps as (first index, step) per eight elements, when they all are progressions
        # packs with a dedicated kernel instead of a map (the cube convolutions' weight images: segm_conv3d_k3_cube_pack_multi)
        self._ckeys: Dict[tuple, tuple] = {}                   # key -> (index, offset, shape) inside _cbuf
        self._citems: list = []                                # per pack: (src_off, out_off, cout_w, cin_w, co_stride, flipped)
        self._ctotal = 0
        self._cbuf: Optional[torch.Tensor] = None
        self._cdesc = None                                     # (device descriptor table, blocks)
        self._cready = 0
        self.frozen = False                                    # a captured graph gathers through _dmap into _dbuf: no new packs
        self._dready = 0                                       # packs [0, _dready) of _dkeys were filled by the last refresh

    def attach_flat_grads(self) -> torch.Tensor:
        """`p.grad` of every parameter becomes a window of ONE flat fp32 buffer laid out like `flat32` (what DDP's
        `gradient_as_bucket_view` does per bucket): a data-parallel run all-reduces the single buffer and the optimizer walks
        three flat arrays instead of 286 tensors.  Two ways to fill it per step: leave the windows attached and zero the buffer
        (autograd then accumulates in place: one add launch per parameter), or `release_grads()` before the backward pass and
        `gather_grads()` behind it (autograd assigns fresh tensors, ONE multi-tensor copy moves them: what the trainer does)."""
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat32)
            self.grad_windows = [self.flat_grad[self.offsets[id(p)]:self.offsets[id(p)] + p.numel()].view(p.shape)
                                 for p in self.params]
        for p, w in zip(self.params, self.grad_windows):
            p.grad = w
        return self.flat_grad

    def release_grads(self) -> None:
        """`p.grad = None` for every parameter (reference trainer.py:445): the backward pass ASSIGNS its gradients"""
        for p in self.params:
            p.grad = None

    def point_grads_at_windows(self) -> None:
        for p, w in zip(self.params, self.grad_windows):
            p.grad = w

    def gather_grads(self, first: int = 0, last: Optional[int] = None) -> None:
        """the gradients the backward pass left on the parameters -> their windows of `flat_grad` by one multi-tensor copy
        (a parameter without a gradient: zeros), and `p.grad` points at the windows again.  Replaces 286 in-place accumulation
        launches plus the zero fill of the buffer (profiles/r03_step_v1_step_kernels.txt: 305 fp32 adds, 1.2 ms per step).
        `first` / `last`: only the parameters [first, last) - one segment of an overlapped exchange (trainer.SegmentedExchange);
        `p.grad` is then left alone (the backward pass may still be running)."""
        part = last is not None
        if part:
            dst, src = [], []
            with torch.no_grad():
                for p, w in zip(self.params[first:last], self.grad_windows[first:last]):
                    g = p.grad
                    if g is None:
                        w.zero_()
                    elif g.data_ptr() != w.data_ptr():
                        dst.append(w)
                        src.append(g if g.dtype == torch.float32 else g.float())
                if dst:
                    torch._foreach_copy_(dst, src)
            return
        dst, src = [], []
        with torch.no_grad():
            for p, w in zip(self.params, self.grad_windows):
                g = p.grad
                if g is None:
                    w.zero_()
                elif g.data_ptr() != w.data_ptr():
                    dst.append(w)
                    src.append(g if g.dtype == torch.float32 else g.float())
            if dst:
                torch._foreach_copy_(dst, src)
        for p, w in zip(self.params, self.grad_windows):
            p.grad = w

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
            if self._dmaps:
                if self._dmap is None or self._dmap.numel() != self._dtotal:
                    self._dmap = torch.cat(self._dmaps)
                    self._dbuf = torch.empty(self._dtotal, dtype=self.dtype, device=self.flat16.device)
                    self._dcompact = None
                    if _lib.on_device(self._dbuf):
                        # eight consecutive packed elements = eight channels of one (row, tap): (first index, step) per group
                        # instead of an index per element (segm_gather16 mode 1), when every pack is laid out that way
                        self._dcompact = _ops_raw().gather16_compact_map(self._dmap)
                if _lib.on_device(self._dbuf):                 # every derived pack, one launch
                    if self._dcompact is not None:
                        _ops_raw().gather16(_lib.get_lib(), self.flat16z, self._dcompact, self._dbuf, compact=True)
                    else:
                        _ops_raw().gather16(_lib.get_lib(), self.flat16z, self._dmap, self._dbuf)
                else:
                    torch.index_select(self.flat16z, 0, self._dmap, out=self._dbuf)
                self._dready = len(self._dmaps)
            if self._citems and _lib.on_device(self.flat16):
                if self._cbuf is None or self._cbuf.numel() != self._ctotal:
                    self._cbuf = torch.empty(self._ctotal, dtype=self.dtype, device=self.flat16.device)
                    self._cdesc = _ops_raw().cube_pack_descs(self._citems, self.flat16.device)
                _ops_raw().conv3d_cube_pack_multi(_lib.get_lib(), self.flat16, self._cbuf, *self._cdesc)      # every cube image, one launch
                self._cready = len(self._citems)
        self.fresh = True

    # ---- derived packs ---------------------------------------------------------------------------------------------------------
    def _window_offset(self, w: torch.Tensor) -> Optional[int]:
        """element offset of `w` inside flat16 when it is a (possibly strided) window of it"""
        if w.dtype != self.dtype or w.device != self.flat16.device:
            return None
        d = w.data_ptr() - self.flat16.data_ptr()
        if d < 0 or d >= 2 * self.flat16.numel() or w.untyped_storage().data_ptr() != self.flat16z.untyped_storage().data_ptr():
            return None
        return d // 2

    def derived_cube_image(self, w: torch.Tensor, flipped: bool, fn):
        """The cube convolutions' weight image of a window `w` (Cout, Cin, 3, 3, 3) of the step's 16-bit copies (a whole weight or a
        channel slice of one): like `derived`, but filled by the dedicated pack launch instead of the generic gather.  `fn(w)`
        computes the image directly (first call, and whenever the window is not one the pack kernel takes)."""
        off = self._window_offset(w)
        ok = (off is not None and self.fresh and _lib.on_device(w) and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and
              tuple(w.stride()[1:]) == (27, 9, 3, 1) and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and w.stride(0) < (1 << 31))
        if not ok:
            return self.derived(w, ("conv3d_cube", bool(flipped)), fn)
        key = (bool(flipped), off, tuple(w.shape), w.stride(0))
        hit = self._ckeys.get(key)
        if hit is not None:
            idx, o, n = hit
            return self._cbuf[o:o + n] if idx < self._cready else fn(w)
        out = fn(w)
        if self.frozen:
            return out
        n = w.shape[0] * w.shape[1] * 27
        self._ckeys[key] = (len(self._citems), self._ctotal, n)
        self._citems.append((off, self._ctotal, w.shape[0], w.shape[1], w.stride(0), bool(flipped)))
        self._ctotal += n
        return out

    def derived(self, w: torch.Tensor, tag, fn):
        """`fn(w)` for a window `w` of the step's 16-bit copies, where `fn` only RE-ARRANGES elements (slice, permute, flip,
        transpose, reshape, zero padding, contiguous).  The first call evaluates `fn` directly and records WHERE every element
        of the result comes from (by running `fn` on the window's element indices); from the next `refresh()` on the pack is a
        view of one buffer that a single gather launch fills for all registered packs."""
        off = self._window_offset(w)
        if off is None or not self.fresh:
            return fn(w)
        key = (tag, off, tuple(w.shape), tuple(w.stride()))
        hit = self._dkeys.get(key)
        if hit is not None:
            idx, o, shape = hit
            if idx < self._dready:
                n = 1
                for d_ in shape:
                    n *= d_
                return self._dbuf[o:o + n].view(shape)
            return fn(w)                                       # registered after the last refresh: direct once more
        out = fn(w)
        if self.frozen:                                        # a HIP graph replays the gather with today's map and buffer: registering a
            return out                                         # pack would re-create both under it (trainer.GraphedStep); stay per-call
        if self._iota is None:
            self._iota = torch.arange(8, 8 + self.flat16.numel(), dtype=torch.int32, device=self.flat16.device)
        m = fn(self._iota.as_strided(w.size(), w.stride(), off))
        if m.dtype != torch.int32 or tuple(m.shape) != tuple(out.shape):
            return out                                         # not a pure re-arrangement: stays a per-call computation
        m = m.reshape(-1)
        pad = (-m.numel()) % 8                                 # 16-byte aligned packs (MFMA operand loads)
        if pad:
            m = torch.cat([m, m.new_zeros(pad)])
        self._dkeys[key] = (len(self._dmaps), self._dtotal, tuple(out.shape))
        self._dmaps.append(m)
        self._dtotal += m.numel()
        return out


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


def packed(w: torch.Tensor, tag, fn):
    """A re-arranged copy `fn(w)` of a weight (packed / padded / transposed layouts the kernels read): inside a bank step and for
    a window of the bank's 16-bit copies it comes out of the bank's one-launch gather (`ParamBank.derived`), else `fn(w)`."""
    if _ACTIVE is not None:
        return _ACTIVE.derived(w, tag, fn)
    return fn(w)


def packed_cube_image(w: torch.Tensor, flipped: bool, fn):
    """the cube convolutions' weight image of `w`: inside a bank step from the bank's dedicated pack launch, else `fn(w)`"""
    if _ACTIVE is not None:
        return _ACTIVE.derived_cube_image(w, flipped, fn)
    return fn(w)


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
