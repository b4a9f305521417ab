"""Dense 3x3x3 (stride 1, "same") convolutions of the SegMamba stem: a dispatcher over the library's own MFMA kernels and
the vendor routes - by a shape table since round 3 (`_table_choice`), by a timing run on request (SEGM_CONV_AUTOTUNE=1).

SURVEY.md §7 step 6: the stem started on MIOpen.  Profiling (profiles/r01_probe_convs.log,
profiles/r01_bench_step_kernels_v3.txt) showed that MIOpen's bf16 3-D solvers are very uneven on these shapes:

  * forward at 48 channels runs at 200-260 TF/s, but 96 -> 96 @64^3 only at 54 TF/s;
  * backward-data for 48 -> 48 @128^3 takes 12.5 ms while the *forward* kernel on the same shape takes 2.5 ms;
  * one layer (decoder2.conv1, 96 -> 48 @128^3) picks solvers that need 75 ms (data) + 530 ms (weights) - more than
    half of a whole training step.

Every quantity below is mathematically the same convolution, only routed to a different library call:

  fwd    y  = conv(x, W)                      or  sum over 48-channel input blocks / concat over output blocks
                                              or  segm_conv3d_k3_fwd, the library's own NCDHW MFMA kernel (per 48-channel
                                                  input block; csrc/conv3d_fwd.hip)
  dgrad  dx = conv_bwd_data(dy, W)            or  conv(dy, flip(W)^T)         (a forward convolution, any of the above)
  wgrad  dW = conv_bwd_weight(x, dy)          or  per input/output channel block
                                              or  segm_conv3d_k3_wgrad, the library's own MFMA kernel

Rounds 1 - 2 timed every candidate the first time a (kind, shapes, dtype, device, strides) was seen and cached the fastest
(what MIOpen's own "find" does, one level up).  The winners were the same on every box (profiles/r01_conv_autotune_v2.log,
r02_bench_variants.log): the library's kernels for every layer of width >= 16, the vendor GEMM route for the 8^3 bottleneck
layers.  That outcome is now the routing table; with SEGM_CONV_AUTOTUNE=1 the timing run is back (SEGM_CONV_VERBOSE=1 prints
its timings).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import linear

_BLOCK = 48                      # channel block of the library's kernels (and of MIOpen's fast 3-D bf16 solvers)
_cache: Dict[tuple, int] = {}
# Routing is a TABLE by default (round 3): the winners of the round-2 timing runs as a rule of the shape (`_table_choice`), so that
# every rank, every run and every captured graph takes the same kernels and results are reproducible run to run.
# SEGM_CONV_AUTOTUNE=1 brings the timing-based choice back (to re-derive the table on new hardware; it synchronises the device
# inside autograd, may differ between processes, and cannot run under graph capture).
_TUNE = os.environ.get("SEGM_CONV_AUTOTUNE", "0") == "1"
# cat(up, skip) convolutions as one autograd node with the later parts added in place (_ConvSameCat, linear._PointwiseCat).  Timed in
# round 3: no gain on the MI355X (66.2 ms per step either way), so it stays opt-in.
_CAT_FUSED = os.environ.get("SEGM_CONV_CAT_FUSED", "1") == "1"     # round 5: default on (with the statistics epilogue the cat layers feed their InstanceNorm too; profiles/r05_inorm_epilogue_step.log)
# round 5: the chained kernels' storing K part sums {count, y, y^2} of what it writes; the InstanceNorm behind the convolution
# merges those partials instead of reading the volume again (csrc/conv3d_fwd.hip STATS; measured free on the convolution side,
# profiles/r05_inorm_epilogue.log).  SEGM_CONV_STATS=0: every InstanceNorm makes its own statistics pass (A/B).
_STATS = os.environ.get("SEGM_CONV_STATS", "1") == "1"
# round 5: conv1 + the 1x1x1 projection of a residual block's input as one node (_ResFront): the two data gradients of the input are
# summed by the 3x3x3 kernel's accumulate mode instead of an element-wise add.  SEGM_RES_FRONT=0: two nodes (A/B).
_RES_FRONT = os.environ.get("SEGM_RES_FRONT", "1") == "1"


# round 6 (late): the wide layers of the 8^3 / 16^3 / 32^3 levels on segm_conv3d_k3_cube_fwd (csrc/conv3d_cube.hip) - forward and data
# gradient; 2.5 - 6x over the row kernels / the vendor route there (profiles/r06_conv_cube_v2.txt).  SEGM_CONV_CUBE=0: the round-5 routing.
_CUBE = os.environ.get("SEGM_CONV_CUBE", "1") == "1"
_CUBE_MAX_WIDTH = int(os.environ.get("SEGM_CONV_CUBE_MAX_WIDTH", "32"))
# ... and the weight gradients of the 8^3 level and of the 384-channel layers at 16^3 on segm_conv3d_k3_cube_wgrad (dY cube and X halo
# cube in LDS, a wave = a 16 x 16 (co, ci) tile x 27 taps): 2.1 - 2.6x over the vendor route at 8^3 - the last MIOpen convolution of the
# step -, 1.7 - 1.8x over the row kernel at 384 -> 384 @16^3; behind it below 384 channels and at 32^3 (profiles/r06_conv_cube_wgrad_v5.txt)
_CUBE_WGRAD = os.environ.get("SEGM_CONV_CUBE_WGRAD", "1") == "1"
_CUBE_WGRAD_MAX_WIDTH = int(os.environ.get("SEGM_CONV_CUBE_WGRAD_MAX_WIDTH", "16"))


def _time(fn: Callable[[], torch.Tensor], reps: int = 3) -> float:
    """median of `reps` timed calls after one warm-up call (close candidates differ by a few per cent)"""
    fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    return sorted(times)[len(times) // 2]


def _key(kind, x, w, *flags) -> tuple:
    """cache key of one routing decision: everything a candidate's speed (or applicability) depends on - shapes, dtype,
    the device, and the operand strides (NCDHW vs channels-last views take different MIOpen solvers)"""
    return (kind, tuple(x.shape), tuple(w.shape), x.dtype, x.device.index, tuple(x.stride()), tuple(w.stride())) + flags


def _table_variant(width: int):
    """the library forward kernel's (chain, pitch48, chain32) flags for a volume of this width (profiles/r02_bench_variants.log,
    r02_conv_stride_pad.log, r04_conv_chain_final.log): chained K parts everywhere; 64-wide blocks with unpadded LDS rows at 128^3;
    the 32-wide-block variant (two workgroups per CU) for 64^3 and below - since round 4, when its storing wave stopped staging
    the halo columns, it is level at 48 -> 48 @64^3 (0.074 ms both) and ahead at 96 -> 96 @64^3 (0.315 vs 0.334 ms)"""
    return (True, True, False) if width >= 128 else (False, False, True)


def _is_lib(v) -> bool:
    """a routing variant that is a library kernel: the row kernels' flag tuple, or "cube" """
    return isinstance(v, tuple) or v == "cube"


def _table_choice(kind: str, width: int, variants) -> int:
    """index of the routing the table prescribes among `variants` (None = a vendor route, tuple = library kernel flags, "mfma" =
    the library's weight-gradient kernel).  Measured on MI355X (profiles/r02_bench_variants.log): the library's kernels win every
    3x3x3 layer of width >= 16 - forward, data gradient (as a forward convolution with flipped, transposed weights) and weight
    gradient; the 8^3 bottleneck layers (768 / 384 channels, 0.1 ms each) stay on the vendor GEMM route, where they are
    weight-bandwidth-bound GEMMs."""
    if kind == "wgrad":
        if "cube" in variants:                           # only offered where it is the table's choice (_wgrad)
            return variants.index("cube")
        return variants.index("mfma") if width >= 16 and "mfma" in variants else 0
    if "cube" in variants:                               # only offered where it is the table's choice (_cube_ok)
        return variants.index("cube")
    if width >= 16:
        want = _table_variant(width)
        if want in variants:
            return variants.index(want)
        lib_routes = [i for i, v in enumerate(variants) if isinstance(v, tuple)]
        if lib_routes:
            return lib_routes[0]
    return 0


def _pick(key: tuple, cands: List[Callable[[], torch.Tensor]], variants=None, width: int = 0, box: list = None) -> torch.Tensor:
    """`box`: the statistics the candidates append to - emptied before the launch whose result is returned, so that what it
    holds afterwards belongs to that launch and not to a timed candidate"""
    if len(cands) == 1 or not torch.cuda.is_available():
        return cands[0]()
    if not _TUNE:
        return cands[_table_choice(key[0], width, variants) if variants is not None else 0]()
    i = _cache.get(key)
    if i is None:
        times = []
        for c in cands:
            try:
                times.append(_time(c))
            except RuntimeError as e:
                # only "this candidate does not take this shape" is a score (the library's status -2 / -4, MIOpen's
                # "no solver"); a HIP launch failure or an out-of-memory error must surface, not be timed as infinity
                msg = str(e)
                if "HIP error" in msg or "out of memory" in msg:
                    raise
                times.append(float("inf"))
        i = min(range(len(cands)), key=lambda j: times[j])
        _cache[key] = i
        if os.environ.get("SEGM_CONV_VERBOSE"):
            print(f"[conv3d autotune] {key}: " + ", ".join(f"{t:.2f} ms" for t in times) + f" -> {i}", flush=True)
    if box is not None:
        box.clear()
    return cands[i]()


def _blocks(c: int) -> List[slice]:
    return [slice(i, min(i + _BLOCK, c)) for i in range(0, c, _BLOCK)]


def _fwd_native(x, w, pad):
    return F.conv3d(x, w, None, 1, pad)


def _hip_fwd_ok(x, w) -> bool:
    from . import ops_raw
    return w.shape[2:] == (3, 3, 3) and (w.shape[1] % _BLOCK == 0 or w.shape[1] < _BLOCK) and w.shape[0] % 16 == 0 \
        and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and \
        ops_raw.conv3d_k3_fwd_supported(x[:, :_BLOCK], w.shape[0])


def _packed_block(w, ib, flipped, dtype):
    """the library kernel's register layout of one 48-channel input block of a weight: (Cout, 3, 3, 3, 48) in the activations'
    dtype; `flipped`: the block of the data-gradient weight flip(W)^T.  A re-arrangement of the step's 16-bit copy: inside a
    bank step it is a view of the bank's gather buffer (param_bank.packed), not a per-call copy chain."""
    from . import ops_raw
    from .param_bank import packed
    if w.dtype != dtype:
        return ops_raw.pack_conv3d_weight((w.flip(2, 3, 4).transpose(0, 1) if flipped else w)[:, ib], dtype)
    if flipped:
        return packed(w, ("conv3d_k3_dgrad", ib.start, ib.stop),
                      lambda t: ops_raw.pack_conv3d_weight(t.flip(2, 3, 4).transpose(0, 1)[:, ib], t.dtype))
    return packed(w, ("conv3d_k3_fwd", ib.start, ib.stop), lambda t: ops_raw.pack_conv3d_weight(t[:, ib], t.dtype))


def _fwd_hip(x, w, pad, bias=None, chain=False, pitch48=False, chain32=False, into=None, flipped=False, stats_box=None):
    """segm_conv3d_k3_fwd per 48-channel input block (the kernel keeps one block's weights in registers).  With
    Cout % 48 == 0 the later blocks accumulate into the first block's output in place; `chain` picks the kernel whose K
    parts are pipelined (csrc/conv3d_fwd.hip, variant 1).  `into`: an existing result every block is added to (the next part
    of a concatenated input; Cout % 48 == 0, no bias).  `stats_box` (a list): the LAST block's launch - the one that stores the
    finished values - also sums the InstanceNorm statistics of the result and appends the partials (kernels with that epilogue only)."""
    from . import lib as L, ops_raw
    hip = L.get_lib()
    cout, cin = (w.shape[1], w.shape[0]) if flipped else (w.shape[0], w.shape[1])     # flipped: w is the forward weight
    inplace = cout % _BLOCK == 0
    out = into
    blocks = _blocks(cin)
    for i, ib in enumerate(blocks):
        wp = _packed_block(w, ib, flipped, x.dtype)
        if inplace:
            last = stats_box is not None and i + 1 == len(blocks) and ((chain and pitch48) or chain32)
            out = ops_raw.conv3d_k3_fwd(hip, x[:, ib], wp, bias if i == 0 else None, out=out, accumulate=i > 0 or into is not None,
                                        chain=chain, pitch48=pitch48, chain32=chain32, want_stats=last)
            if last:
                out, st = out
                if st is not None:
                    stats_box.append(st)
        else:
            y = ops_raw.conv3d_k3_fwd(hip, x[:, ib], wp, bias if i == 0 else None)
            out = y if out is None else out + y
    return out


def _cube_ok(x, cout: int, cin_w: int) -> bool:
    """the cube kernel takes this convolution AND the table prefers it: every supported layer of width <= 16, at 32^3 the layers from
    96 -> 192 channels up (96 -> 96 is level with the row kernel, which also carries the statistics epilogue); with the tuner on,
    wherever it is supported"""
    from . import ops_raw
    if not _CUBE or not ops_raw.conv3d_cube_supported(x, cout) or x.shape[1] != cin_w:
        return False
    if _TUNE:
        return True
    width = x.shape[4]
    return width <= min(16, _CUBE_MAX_WIDTH) or (width <= _CUBE_MAX_WIDTH and width <= 32 and x.shape[1] * cout >= 96 * 192)


def _fwd_cube(x, w, bias=None, into=None, flipped=False, stats_box=None):
    """segm_conv3d_k3_cube_fwd: the whole contraction in one launch (+ its reduction); `flipped`: w is the forward weight and x
    the output gradient (the data gradient); `into`: an existing result the launch adds to; `stats_box`: the InstanceNorm partials of
    what the launch stores are appended"""
    from . import lib as L, ops_raw
    from .param_bank import packed_cube_image
    hip = L.get_lib()
    cout = w.shape[1] if flipped else w.shape[0]
    if w.dtype != x.dtype:
        img = ops_raw.conv3d_cube_weight_image(hip, w, flipped, x.dtype)
    else:                                                  # inside a bank step: one of the images the step's pack launch refreshes
        img = packed_cube_image(w, flipped, lambda t: ops_raw.conv3d_cube_weight_image(hip, t, flipped))
    res = ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias, out=into, accumulate=into is not None, want_stats=stats_box is not None)
    if stats_box is not None:
        res, st = res
        stats_box.append(st)
    return res


def _fwd_lib(v, x, w, pad, bias=None, into=None, stats_box=None, flipped=False):
    """the library kernel variant `v` (a row-kernel flag tuple or "cube") on one part of a convolution"""
    if v == "cube":
        return _fwd_cube(x, w, bias, into=into, flipped=flipped, stats_box=stats_box)
    return _fwd_hip(x, w, pad, bias, *v, into=into, flipped=flipped, stats_box=stats_box)


def _hip_chain_ok(w) -> bool:
    """The chained-K-parts kernel (eight-wave layout) as an extra autotune candidate: on MI355X it beats the
    reduce-per-row kernel by 5 - 16 % on the 128^3 / 64^3 layers and loses ~4 % at 32^3 (profiles/r01_conv_chain_ab.log), so the
    tuner decides per shape.  SEGM_CONV_FWD_CHAIN=0 removes the candidate."""
    return w.shape[0] % _BLOCK == 0 and os.environ.get("SEGM_CONV_FWD_CHAIN", "1") != "0"


def _hip_untimed_ok() -> bool:
    """Two further variants of the chained kernel - unpadded LDS rows, 32-wide x blocks with two workgroups per CU.  Measured in
    round 2 (profiles/r02_bench_variants.log): the 32-wide variant wins every 32^3 / 16^3 layer (96 -> 96 @32^3 0.15 -> 0.11 ms,
    192 -> 192 @32^3 0.43 -> 0.27 ms, 384 -> 384 @16^3 0.41 -> 0.31 ms, where MIOpen was the previous winner) and the step drops from
    79.2 to 77.9 ms, so both are candidates by default; SEGM_CONV_FWD_UNTIMED=0 removes them."""
    return os.environ.get("SEGM_CONV_FWD_UNTIMED", "1") == "1"


def _fwd_blocked(x, w, pad):
    outs = []
    for ob in _blocks(w.shape[0]):
        acc = None
        for ib in _blocks(w.shape[1]):
            y = F.conv3d(x[:, ib], w[ob, ib], None, 1, pad)
            acc = y if acc is None else acc + y
        outs.append(acc)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


def _dgrad_native(dy, w, x, pad):
    return torch.ops.aten.convolution_backward(dy, x, w, None,
                                                [1, 1, 1], [pad] * 3, [1, 1, 1], False, [0, 0, 0], 1,
                                                [True, False, False])[0]


def _flipT(w):
    return w.flip(2, 3, 4).transpose(0, 1).contiguous()


def _dgrad_as_fwd(dy, w, x, pad):
    return F.conv3d(dy, _flipT(w), None, 1, pad)                 # valid for stride 1 and pad == k // 2


def _dgrad_as_fwd_blocked(dy, w, x, pad):
    return _fwd_blocked(dy, _flipT(w), pad)


def _dgrad_hip(dy, w, x, pad, chain=False, pitch48=False, chain32=False, into=None):
    return _fwd_hip(dy, w, pad, None, chain, pitch48, chain32, into=into, flipped=True)


def _wgrad_native(x, dy, w, pad):
    return torch.ops.aten.convolution_backward(dy, x, w, None,
                                                [1, 1, 1], [pad] * 3, [1, 1, 1], False, [0, 0, 0], 1,
                                                [False, True, False])[1]


def _wgrad_blocked(x, dy, w, pad):
    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
    for ob in _blocks(w.shape[0]):
        for ib in _blocks(w.shape[1]):
            dw[ob, ib] = _wgrad_native(x[:, ib].contiguous(), dy[:, ob].contiguous(), w[ob, ib].contiguous(), pad)
    return dw


def _wgrad_mfma(x, dy, w, pad, out_dtype=None):
    """segm_conv3d_k3_wgrad: the hand-written MFMA weight-gradient kernel (csrc/conv3d_wgrad.hip); the result in the dtype
    of the master weight (fp32 under autocast: the kernel's accumulators, not rounded)."""
    from . import lib as L, ops_raw
    if not ops_raw.conv3d_k3_wgrad_supported(x, dy):
        x = x.contiguous()                       # e.g. the permuted (channel-last) outputs of the Mamba encoder
    return ops_raw.conv3d_k3_wgrad(L.get_lib(), x, dy, out_dtype or w.dtype)


def _cube_wgrad_ok(x, dy, w) -> bool:
    """segm_conv3d_k3_cube_wgrad takes the layer AND the table prefers it (8^3: every wide layer; 16^3: from 384 x 384 channels; with
    the tuner on, wherever it is supported)"""
    from . import ops_raw
    if not _CUBE_WGRAD or w.shape[2:] != (3, 3, 3) or not ops_raw.conv3d_cube_wgrad_supported(x, dy):
        return False
    if _TUNE:
        return True
    width, prod = x.shape[4], x.shape[1] * dy.shape[1]
    return width <= _CUBE_WGRAD_MAX_WIDTH and ((width <= 8 and prod >= 96 * 192) or (width <= 16 and prod >= 384 * 384))


def _wgrad_cube(x, dy, out_dtype):
    from . import lib as L, ops_raw
    return ops_raw.conv3d_k3_cube_wgrad(L.get_lib(), x, dy, out_dtype if out_dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32)


def _mfma_wgrad_ok(x, dy, w) -> bool:
    from . import ops_raw
    if w.shape[2:] != (3, 3, 3) or w.dtype not in (torch.bfloat16, torch.float16, torch.float32) or \
            x.dtype not in (torch.bfloat16, torch.float16):
        return False
    return (x.shape[1] % 48 == 0 or x.shape[1] < 48) and dy.shape[1] % 48 == 0 and x.shape[4] % 8 == 0 and \
        x.shape[0] == dy.shape[0] and ops_raw.conv3d_k3_wgrad_spans_fit(x, dy)


_HIP_VARIANTS = ((False, False, False), (True, False, False), (True, True, False), (False, False, True))   # (chain, pitch48, chain32)


def _fwd_candidates(x, w, bias, pad, stats_box=None):
    """-> (key, candidates, variant per candidate): the mathematically identical routings of one forward convolution; variant =
    the library kernel's (chain, pitch48, chain32) flags, None for the vendor routes"""
    hip = _hip_fwd_ok(x, w)
    chain = hip and _hip_chain_ok(w)
    key = _key("fwd", x, w, hip, chain, _hip_untimed_ok(), _CUBE)

    def with_bias(y):
        return y if bias is None else y + bias.view(1, -1, 1, 1, 1)

    cands, variants = [lambda: F.conv3d(x, w, bias, 1, pad)], [None]
    if max(w.shape[0], w.shape[1]) > _BLOCK and w.shape[0] % _BLOCK == 0 and w.shape[1] % _BLOCK == 0:
        cands.append(lambda: with_bias(_fwd_blocked(x, w, pad)))
        variants.append(None)
    n = (1 if hip else 0) + (1 if chain else 0) + (2 if chain and _hip_untimed_ok() else 0)
    for v in _HIP_VARIANTS[:n]:                          # bias fused into the kernel's epilogue
        cands.append(lambda v=v: _fwd_hip(x, w, pad, bias, *v, stats_box=stats_box))
        variants.append(v)
    if w.shape[2:] == (3, 3, 3) and w.dtype == x.dtype and _cube_ok(x, w.shape[0], w.shape[1]):
        cands.append(lambda: _fwd_cube(x, w, bias, stats_box=stats_box))
        variants.append("cube")
    return key, cands, variants


def _pick_was_hip(key, cands, variants, width: int = 0) -> bool:
    """did `_pick` return the result of a library-kernel candidate (only then are statistics in the box those of the result)"""
    if len(cands) == 1 or not torch.cuda.is_available():
        return _is_lib(variants[0])
    if not _TUNE:
        return _is_lib(variants[_table_choice(key[0], width, variants)])
    i = _cache.get(key)
    return i is not None and _is_lib(variants[i])


def _tuned_variant(key, cands, variants, width: int = 0):
    """the library-kernel variant routing gives this key (table, or what the tuner has already picked), or None (a vendor route)"""
    if len(cands) == 1 or not torch.cuda.is_available():
        return None
    if not _TUNE:
        v = variants[_table_choice(key[0], width, variants)]
        return v if _is_lib(v) else None
    i = _cache.get(key)
    return variants[i] if i is not None and _is_lib(variants[i]) else None


def _dgrad(dy, w, x, pad, into=None):
    """`into`: a gradient of x that already exists (another consumer's, owned by the caller) - the result is added to it, in place
    by the library kernel when the shape routes to one (`accumulate`), by an ordinary add otherwise"""
    cands, variants = [lambda: _dgrad_native(dy, w, x, pad), lambda: _dgrad_as_fwd(dy, w, x, pad)], [None, None]
    if max(w.shape[0], w.shape[1]) > _BLOCK and w.shape[0] % _BLOCK == 0 and w.shape[1] % _BLOCK == 0:
        cands.append(lambda: _dgrad_as_fwd_blocked(dy, w, x, pad))
        variants.append(None)
    hip = _hip_fwd_ok(dy, w.transpose(0, 1))
    chain = hip and _hip_chain_ok(w.transpose(0, 1))
    n = (1 if hip else 0) + (1 if chain else 0) + (2 if chain and _hip_untimed_ok() else 0)
    for v in _HIP_VARIANTS[:n]:
        cands.append(lambda v=v: _dgrad_hip(dy, w, x, pad, *v))
        variants.append(v)
    if w.shape[2:] == (3, 3, 3) and w.dtype == dy.dtype and _cube_ok(dy, w.shape[1], w.shape[0]):
        cands.append(lambda: _fwd_cube(dy, w, flipped=True))
        variants.append("cube")
    key = _key("dgrad", dy, w, hip, chain, _hip_untimed_ok(), _CUBE)
    if into is not None:
        from . import ops_raw
        v = _tuned_variant(key, cands, variants, dy.shape[4]) if w.shape[1] % _BLOCK == 0 and ops_raw.channel_dense(into) else None
        if v is not None:
            return _fwd_lib(v, dy, w, pad, into=into, flipped=True)
        return into.add_(_pick(key, cands, variants, dy.shape[4]))
    return _pick(key, cands, variants, dy.shape[4])


def _wgrad(x, dy, w, pad, w_dtype):
    cands, variants = [lambda: _wgrad_native(x, dy, w, pad)], [None]
    if max(w.shape[0], w.shape[1]) > _BLOCK and w.shape[0] % _BLOCK == 0 and w.shape[1] % _BLOCK == 0:
        cands.append(lambda: _wgrad_blocked(x, dy, w, pad))
        variants.append(None)
    mfma = _mfma_wgrad_ok(x, dy, w)
    if mfma:
        cands.append(lambda: _wgrad_mfma(x, dy, w, pad, w_dtype))
        variants.append("mfma")
    cube = _cube_wgrad_ok(x, dy, w)
    if cube:
        cands.append(lambda: _wgrad_cube(x, dy, w_dtype))
        variants.append("cube")
    return _pick(_key("wgrad", x, w, mfma, w_dtype, cube), cands, variants, x.shape[4]).to(w_dtype)


class _ConvSame(torch.autograd.Function):
    """stride-1 "same" convolution, odd kernel; x already in the compute dtype, w / bias in any dtype (fp32 masters under autocast)."""

    @staticmethod
    def forward(ctx, x, w, bias, want_stats=False):
        from .linear import _masters
        w, bias = _masters(ctx, x, w, bias)              # fp32 masters -> the step's 16-bit copies; gradients go back in fp32
        ctx.save_for_backward(x, w)
        box = [] if want_stats else None
        key, cands, variants = _fwd_candidates(x, w, bias, w.shape[2] // 2, box)
        out = _pick(key, cands, variants, x.shape[4], box)
        ctx.with_stats = bool(want_stats)
        if not want_stats:
            return out
        # the tuner empties the box before the launch whose result is returned: what it holds now came from that launch (and a
        # winning variant without the statistics epilogue leaves it empty)
        stats = box[-1] if box and _pick_was_hip(key, cands, variants, x.shape[4]) else out.new_empty(0, dtype=torch.float32)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dy, *_):
        x, w = ctx.saved_tensors
        pad = w.shape[2] // 2
        from . import ops_raw
        if not ops_raw.channel_dense(dy):                # the kernels take strides; a padded channel stride is not copied
            dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(dy, w, x, pad)
        if ctx.needs_input_grad[1]:
            dw = _wgrad(x, dy, w, pad, ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = linear.bias_grad(dy).to(ctx.b_dtype)
        return dx, dw, db, None


class _ConvSameCat(torch.autograd.Function):
    """conv3d_same(cat(xs, 1), w) without the concatenation AND without the adds: the convolution is linear in its input
    channels; the first part is a plain convolution with w[:, :c0], every later part is added in place by the library kernel the
    tuner picked for its shape (`accumulate`) - or, when a vendor route won for that shape, through an ordinary add.  One node
    for the whole layer: the weight gradient is assembled by one cat instead of autograd's zero-fill + copy + add per slice."""

    @staticmethod
    def forward(ctx, want_stats, w, *xs):
        from .param_bank import low_precision
        ctx.w_dtype = w.dtype
        w = low_precision(w, xs[0].dtype)
        pad = w.shape[2] // 2
        ctx.save_for_backward(w, *xs)
        out, c0, stats = None, 0, None
        for j, x in enumerate(xs):
            wi = w[:, c0:c0 + x.shape[1]]
            c0 += x.shape[1]
            key, cands, variants = _fwd_candidates(x, wi, None, pad)
            if out is None:
                out = _pick(key, cands, variants, x.shape[4])
                continue
            v = _tuned_variant(key, cands, variants, x.shape[4]) if wi.shape[0] % _BLOCK == 0 else None
            if v is not None:
                box = [] if (want_stats and j + 1 == len(xs)) else None      # the launch that stores the finished values
                out = _fwd_lib(v, x, wi, pad, into=out, stats_box=box)
                stats = box[-1] if box else None
            else:                                        # not tuned yet (this call does it) or a vendor route won
                out = out + _pick(key, cands, variants, x.shape[4])
        if not want_stats:
            return out
        stats = stats if stats is not None else out.new_empty(0, dtype=torch.float32)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dy, *_):
        w, *xs = ctx.saved_tensors
        pad = w.shape[2] // 2
        from . import ops_raw
        if not ops_raw.channel_dense(dy):
            dy = dy.contiguous()
        dxs, dws, c0 = [], [], 0
        for i, x in enumerate(xs):
            wi = w[:, c0:c0 + x.shape[1]]
            c0 += x.shape[1]
            dxs.append(_dgrad(dy, wi, x, pad) if ctx.needs_input_grad[2 + i] else None)
            if ctx.needs_input_grad[1]:
                dws.append(_wgrad(x, dy, wi, pad, ctx.w_dtype))
        dw = torch.cat(dws, dim=1) if ctx.needs_input_grad[1] else None
        return (None, dw, *dxs)


class _ResFront(torch.autograd.Function):
    """The two convolutions a residual block with a projection skip applies to its input - conv1 (3x3x3) and conv3 (1x1x1),
    dynunet_block.py:93-104 of the reference - on cat(xs, 1), as ONE node: each part then has ONE data gradient, written by the
    1x1x1 kernel and added to in place by the 3x3x3 kernel, where two nodes leave autograd an element-wise add per part (three
    passes over a 128^3 volume each).  Forward arithmetic = _ConvSameCat + linear._PointwiseCat.  -> (y1, stats of y1 or an empty
    tensor, y3)."""

    @staticmethod
    def forward(ctx, want_stats, w1, w3, *xs):
        from .param_bank import low_precision
        ctx.w_dtypes = (w1.dtype, w3.dtype)
        dt = xs[0].dtype
        w1, w3 = low_precision(w1, dt), low_precision(w3, dt)
        pad = w1.shape[2] // 2
        ctx.save_for_backward(w1, w3, *xs)
        y1, y3, c0, stats = None, None, 0, None
        for j, x in enumerate(xs):
            c1 = c0 + x.shape[1]
            w1i, w3i, xf = w1[:, c0:c1], w3[:, c0:c1], x.flatten(2)
            c0 = c1
            last = want_stats and j + 1 == len(xs)
            box = [] if last else None
            if y1 is None:
                key, cands, variants = _fwd_candidates(x, w1i, None, pad, box)
                y1 = _pick(key, cands, variants, x.shape[4], box)
                if last and box and _pick_was_hip(key, cands, variants, x.shape[4]):
                    stats = box[-1]
                y3 = linear._pw_hip(w3i, xf, None)
                if y3 is None:
                    y3 = linear._bmm_w(w3i, xf)
                continue
            key, cands, variants = _fwd_candidates(x, w1i, None, pad)
            v = _tuned_variant(key, cands, variants, x.shape[4]) if w1i.shape[0] % _BLOCK == 0 else None
            if v is not None:
                y1 = _fwd_lib(v, x, w1i, pad, into=y1, stats_box=box)
                stats = box[-1] if box else None
            else:
                y1 = y1 + _pick(key, cands, variants, x.shape[4])
            y = linear._pw_hip(w3i, xf, None, into=y3)
            y3 = y if y is not None else y3 + linear._bmm_w(w3i, xf)
        stats = stats if stats is not None else y1.new_empty(0, dtype=torch.float32)
        ctx.mark_non_differentiable(stats)
        return y1, stats, y3.view(y1.shape)

    @staticmethod
    def backward(ctx, dy1, _, dy3):
        w1, w3, *xs = ctx.saved_tensors
        pad = w1.shape[2] // 2
        from . import ops_raw
        if not ops_raw.channel_dense(dy1):
            dy1 = dy1.contiguous()
        dy3 = dy3.flatten(2) if ops_raw.channel_dense(dy3) else dy3.contiguous().flatten(2)
        dxs, dw1, dw3, c0 = [], [], [], 0
        for i, x in enumerate(xs):
            c1 = c0 + x.shape[1]
            w1i, w3i = w1[:, c0:c1], w3[:, c0:c1]
            c0 = c1
            dx = None
            if ctx.needs_input_grad[3 + i]:
                dx = linear._pw_hip(w3i.t(), dy3, None)
                if dx is None:
                    dx = linear._bmm_w(w3i.t(), dy3)
                dx = _dgrad(dy1, w1i, x, pad, into=dx.view(x.shape))
            dxs.append(dx)
            if ctx.needs_input_grad[1]:
                dw1.append(_wgrad(x, dy1, w1i, pad, ctx.w_dtypes[0]))
            if ctx.needs_input_grad[2]:
                dw3.append(linear.nt_matmul_rows(dy3, x.flatten(2)))
        g1 = torch.cat(dw1, dim=1) if ctx.needs_input_grad[1] else None
        g3 = torch.cat(dw3, dim=1).to(ctx.w_dtypes[1]) if ctx.needs_input_grad[2] else None
        return (None, g1, g3, *dxs)


def res_front(xs: Tuple[torch.Tensor, ...], w1: torch.Tensor, w3: torch.Tensor, want_stats: bool = False):
    """(conv3d_same(cat(xs, 1), w1), its InstanceNorm partials or None, the 1x1x1 convolution of cat(xs, 1) with w3) as one autograd
    node (_ResFront), or None when the parts are not the library kernels' (device, one 16-bit dtype, dense channels, more than 4
    input channels) - the caller then runs the two convolutions on their own."""
    from . import lib as L, ops_raw
    if not (_RES_FRONT and _CAT_FUSED and all(L.on_device(x) for x in xs) and w1.shape[2:] == (3, 3, 3) and w1.shape[1] > 4):
        return None
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        xs = tuple(x.to(dt) for x in xs)
    if xs[0].dtype not in (torch.bfloat16, torch.float16) or any(x.dtype != xs[0].dtype or not ops_raw.channel_dense(x) for x in xs):
        return None
    y1, st, y3 = _ResFront.apply(bool(want_stats and _STATS), w1, w3.reshape(w3.shape[0], w3.shape[1]), *xs)
    return y1, (st if st.numel() else None), y3


def _unpack_stats(res):
    y, st = res
    return y, (st if st.numel() else None)


def conv3d_same(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, want_stats: bool = False):
    """Conv3d(kernel k odd, stride 1, padding k//2).  Follows autocast like `F.conv3d` does.
    want_stats: -> (y, stats); stats = the InstanceNorm partials of y for fused_norm.instance_norm_act(y, ..., stats=stats) when the
    launch that produced y has the statistics epilogue, else None."""
    from . import lib as L
    if not L.on_device(x):
        y = F.conv3d(x, weight, bias, 1, weight.shape[2] // 2)
        return (y, None) if want_stats else y
    if torch.is_autocast_enabled():
        x = x.to(torch.get_autocast_dtype("cuda"))       # the weights stay masters: _ConvSame makes its own copies
    if want_stats and _STATS:
        return _unpack_stats(_ConvSame.apply(x, weight, bias, True))
    y = _ConvSame.apply(x, weight, bias)
    return (y, None) if want_stats else y


def conv3d_same_cat(xs: Tuple[torch.Tensor, ...], weight: torch.Tensor, want_stats: bool = False):
    """conv3d_same(torch.cat(xs, 1), weight) without materialising the concatenation: the convolution is linear in
    its input channels, so it is the sum of convolutions of the parts with the matching weight slices.  (The UNETR
    decoder convolves cat(upsampled, skip), unetr_block.py:82-84; MIOpen's 96 -> 48 @128^3 solver is the 600 ms one.)"""
    from . import lib as L
    if _CAT_FUSED and len(xs) > 1 and all(L.on_device(x) for x in xs):
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            xs = tuple(x.to(dt) for x in xs)
        if all(x.dtype == xs[0].dtype for x in xs):
            if want_stats and _STATS:
                return _unpack_stats(_ConvSameCat.apply(True, weight, *xs))
            y = _ConvSameCat.apply(False, weight, *xs)
            return (y, None) if want_stats else y
    out, c0 = None, 0
    for x in xs:
        c = x.shape[1]
        y = conv3d_same(x, weight[:, c0:c0 + c])
        out = y if out is None else out + y
        c0 += c
    return (out, None) if want_stats else out
