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
  * multi-GPU is data parallel by volume over RCCL, one process per GPU launched by torchrun, batch per GPU 2 (weak scaling).
    Two forms of the same arithmetic (mean of the ranks' gradients, then one identical update on every rank):
      ddp="flat" (default on the GPU)  the parameters, their gradients and momenta live in three flat fp32 arrays
                 (param_bank.py); a step is  [forward + backward]  ->  all-reduce of the flat gradient array
                 (269.7 MB over xGMI)  ->  clip + SGD over the flat arrays.  Eager launches: the array is exchanged in K
                 segments, each on a side stream as soon as the backward pass has produced it (`SegmentedExchange`, K = 4;
                 round 4 - the one call behind the backward pass was fully exposed).  The bracket has fixed addresses, so it
                 can also be captured once as a HIP graph (`GraphedStep`) and replayed - ~1500 kernel launches per step cost the
                 host one `hipGraphLaunch` - with the exchange as ONE call behind the replay (a graph holds no host decisions);
      ddp="torch"  the reference's wrapper - DistributedDataParallel (trainer.py:353-357), bucketed all-reduce overlapped
                 with the backward pass, eager launches.
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
    flat: bool = False             # gradients / momenta in flat arrays next to the bank's parameters (attach_flat_grads)
    world: int = 1                 # ranks averaging their gradients through ONE all-reduce of the flat array (ddp="flat")
    graphed: object = None         # GraphedStep once the forward + backward bracket has been captured
    exchange: object = None        # SegmentedExchange: the flat all-reduce in segments overlapped with the backward pass
    found_inf: object = None       # flat fp16 mode: 1-element tensor the fused step writes (1 = inf / nan gradients, step skipped)


def build_training_state(device, distributed: bool = False, local_rank: int = 0, max_steps: int = 250 * 1000,
                         model: nn.Module | None = None, amp: str | None = None, ddp: str | None = None,
                         flat: bool | None = None, ddp_segments: int | None = None) -> TrainingState:
    """ddp: "flat" | "torch" (see the module docstring; default SEGM_DDP, else "flat" where the parameter bank exists).
    flat: keep gradients / momenta in flat arrays also on one GPU (default: yes with the bank and the fused optimizer).
    ddp_segments: ddp="flat" exchanges the flat gradient array in this many segments, each started as soon as the backward pass
    has produced it (SegmentedExchange; default SEGM_DDP_SEGMENTS, else 4); 1 = one all-reduce behind the backward pass."""
    amp = (amp or os.environ.get("SEGM_AMP", "bf16")).lower()
    if amp not in ("bf16", "fp16"):
        raise ValueError(f"amp must be 'bf16' or 'fp16', got {amp!r}")
    ddp = (ddp or os.environ.get("SEGM_DDP", "flat")).lower()
    if ddp not in ("flat", "torch"):
        raise ValueError(f"ddp must be 'flat' or 'torch', got {ddp!r}")
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
    fused_ok = L.on_device(next(model.parameters())) and os.environ.get("SEGM_FUSED_TRAIN_OPS", "1") != "0"
    if flat is None:
        flat = os.environ.get("SEGM_FLAT_GRADS", "1") == "1"
    flat = bool(flat and bank is not None and fused_ok)
    if flat and amp == "fp16" and not _grad_scaler_internals_ok():
        flat = False                                       # this torch's GradScaler lacks what flat fp16 leans on: the public route
    world = 1
    if distributed and flat and ddp == "flat":
        import torch.distributed as dist
        world = dist.get_world_size()
        dist.broadcast(bank.flat32, 0)                     # what DDP's constructor does: every rank starts from rank 0's weights
        distributed = False                                # no wrapper: the exchange step is one all-reduce in finish_step()
    if distributed:
        # The reference passes find_unused_parameters=True (trainer.py:354-357); every SegMamba parameter takes part in
        # every step (DDP itself reports "did not find any unused parameters"), so the extra per-step graph traversal
        # is dropped.  Gradients live in the communication buckets (no copy in, no copy out), and the buckets are
        # large: xGMI rings are per-link bound, a few 64 MB all-reduces beat many 25 MB ones.  SEGM_DDP_FIND_UNUSED=1
        # restores the reference's flag.
        model = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[local_rank] if device.type == "cuda" else None,
            **DDP_SETTINGS)
    fused = fused_ok
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
    st = TrainingState(model=model, optimizer=opt, scheduler=sched, loss_fn=loss_fn, bank=bank, world=world)
    if flat and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
        bank.attach_flat_grads()
        opt.use_flat(bank)
        st.flat = True
    # SEGM_FORCE_DDP=1: the exchange machinery on ONE rank (hooks, side stream, RCCL on slices) - what a single-GPU box can test
    forced = os.environ.get("SEGM_FORCE_DDP") == "1" and torch.distributed.is_available() and torch.distributed.is_initialized()
    if st.flat and (world > 1 or forced):
        k = int(ddp_segments if ddp_segments is not None else os.environ.get("SEGM_DDP_SEGMENTS", "4"))
        if k > 1:
            st.exchange = SegmentedExchange(bank, world, k)
    if amp == "fp16":
        st.autocast_dtype = torch.float16
        st.scaler = torch.amp.GradScaler(device.type)     # the reference's GradScaler() defaults: 2^16, x2 / 2000 steps, x0.5 on inf
        if st.flat:
            # flat mode keeps the GradScaler as the holder of the scale / growth-tracker tensors; unscale_, the inf check and the
            # skipped step happen inside the fused clip + SGD pass (FusedClipSGD.use_loss_scale), the scale update is the same
            # device-side op scaler.update() runs.  Nothing in the step reads the scale on the host: the bracket stays capturable.
            st.scaler._lazy_init_scale_growth_tracker(device)
            st.found_inf = opt.use_loss_scale(st.scaler._scale)
    return st


def _grad_scaler_internals_ok() -> bool:
    """fp16 in flat mode keeps GradScaler as the holder of the scale / growth-tracker tensors and runs its update op itself
    (`_lazy_init_scale_growth_tracker`, `_scale`, `_growth_tracker`, `torch._amp_update_scale_`: private names of torch 2.x).
    A torch that lacks any of them gets the per-tensor `scale -> unscale_ -> clip -> scaler.step -> update` route instead of an
    AttributeError in the first step.  One difference of the fused route, for extreme values only: its inf check is
    the fp32 sum of squares of the SCALED gradients, so a step is also skipped when that sum overflows although every unscaled
    gradient is finite (|g| * scale > ~1.8e19 somewhere); `GradScaler.unscale_` would keep such a step."""
    try:
        sc = torch.amp.GradScaler("cpu", enabled=True)
        return all(hasattr(sc, n) for n in ("_lazy_init_scale_growth_tracker", "_scale", "_growth_tracker", "get_growth_factor",
                                            "get_backoff_factor", "get_growth_interval")) and hasattr(torch, "_amp_update_scale_")
    except Exception:                                      # noqa: BLE001 - any surprise in a private API: the public route
        return False


def train_step(st: TrainingState, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    if st.flat:
        if st.graphed is not None:
            return st.graphed(image, label)
        loss = forward_backward(st, image, label)
        finish_step(st)
        return loss
    if st.bank is None:
        return _train_step(st, image, label)
    with st.bank.step():                                           # forward and backward see one set of 16-bit parameter copies
        return _train_step(st, image, label)


class SegmentedExchange:
    """The gradient exchange of ddp="flat" in K segments that start while the backward pass is still running - what DDP's
    bucketed all-reduce does for the reference (light_training/trainer.py:353-357), on the flat array.

    The backward pass produces gradients in reverse parameter order (decoder first, stem last).  The flat array is cut at
    parameter boundaries into K contiguous segments of about equal bytes; a post-accumulate hook per parameter counts a
    segment down, and when its last gradient exists the segment is copied into its window of the flat array (main stream), an
    event is recorded and a side stream issues `all_reduce(flat_grad[a:b])` behind it.  finish() makes the main stream wait
    for the side stream (and exchanges what the hooks never completed: parameters without a gradient).  The sums are
    element-wise, so the result equals the one-call exchange bit for bit.  K = 1: the one call behind the backward pass.
    A captured HIP graph cannot hold the hooks' host decisions: while `suspended` (GraphedStep's capture and its replays) the
    hooks do nothing and finish() is the one call.

    Nothing here may hang or mislead the first N > 1 run (round 5):
      * every rank must issue the SAME sequence of collectives.  The order in which segments complete is a property of the
        autograd graph, identical on every rank - but nothing is sent on that assumption: in the FIRST step the hooks only record
        the order, finish() exchanges every segment in index order, and an all-gather compares the recorded orders; only if all
        ranks agree does it become `order` and do the hooks send from the second step on (a mismatch switches EVERY rank to the
        one-call form and says so in `fallback_reason`).  A segment that later completes out of that order stops that rank's hook
        sends BEFORE anything is issued, and finish() sends what the hooks did not in the recorded order - so a rank whose hook
        stopped early still pairs its collectives with the other ranks';
      * a hook that raises (or a send that raises) is caught: the rank stops sending from hooks, finish() completes the step's
        exchange in the recorded order, and a one-element flag all-reduce (MAX) at the end of every finish() tells all ranks;
        from the next step on every rank uses the one-call form (`suspended`).  The step in which it happened is still exact;
      * a collective that never completes is the process group's timeout (bench.py passes 180 s to init_process_group): the job
        dies with a message instead of sitting in a collective until the driver's limit.
    Hooks go only on parameters that require a gradient (register_post_accumulate_grad_hook raises otherwise; a frozen
    parameter's window stays zero and travels with its segment); close() removes them; the side stream lives on the gradients'
    device, not on whatever device is current."""

    def __init__(self, bank, world: int, segments: int):
        self.bank, self.world = bank, world
        self.suspended = False
        params = bank.params
        sizes = [p.numel() for p in params]
        total = sum(sizes)
        k = max(1, min(int(segments), len(params)))
        self.ranges = []                                   # (first parameter, one past the last, element offset a, b)
        i0, acc, a = 0, 0, bank.offsets[id(params[0])]
        for i, n in enumerate(sizes):
            acc += n
            if acc >= total * (len(self.ranges) + 1) / k or i == len(sizes) - 1:
                b = bank.offsets[id(params[i])] + n
                self.ranges.append((i0, i + 1, a, b))
                i0, a = i + 1, b
        self.seg_of = {}
        self.hooked = [0] * len(self.ranges)               # parameters with a hook per segment (those that require a gradient)
        for s, (p0, p1, _, _) in enumerate(self.ranges):
            for i in range(p0, p1):
                self.seg_of[id(params[i])] = s
                self.hooked[s] += int(params[i].requires_grad)
        self.left = [0] * len(self.ranges)
        self.sent = [True] * len(self.ranges)
        dev = bank.flat_grad.device
        self.device = dev
        self.stream = torch.cuda.Stream(device=dev) if bank.flat_grad.is_cuda else None
        self.works = []
        self.exposed_events = []                           # (backward done, exchange done) on the main stream, per step
        self.record_exposed = False
        self.order = None                                  # canonical send order (segment indices), recorded in the first step
        self.this_order = []
        self.steps_done = 0
        self.observed = []                                 # segments in the order their last gradient arrived, this step
        self.failed = None                                 # this rank's first exception (repr), if any
        self.fallback_reason = None                        # why every rank went to the one-call form, if they did
        self._flag = torch.zeros(1, dtype=torch.float32, device=dev)
        self._pending = []                                 # (pinned flag copy, event) of the last steps: read two steps later
        self._hosts = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in params if p.requires_grad]

    def close(self):
        """remove the hooks (a second build_training_state on the same model must not stack them)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def arm(self):
        """before the backward pass of a step"""
        self.left = list(self.hooked)
        self.sent = [False] * len(self.ranges)
        self.works = []
        self.this_order = []
        self.observed = []

    def _hook(self, p):
        if self.suspended or self.failed is not None or all(self.sent):
            return
        try:
            s = self.seg_of[id(p)]
            self.left[s] -= 1
            if self.left[s] == 0 and not self.sent[s]:
                self.observed.append(s)
                if self.order is None:
                    return                                  # first step: record only (finish() exchanges in index order)
                if len(self.this_order) >= len(self.order) or self.order[len(self.this_order)] != s:
                    raise RuntimeError(f"segment {s} completed out of the recorded order {self.order} at position {len(self.this_order)}")
                self._send(s)
        except Exception as e:                              # noqa: BLE001 - never out of an autograd hook: finish() completes the step
            self.failed = f"{type(e).__name__}: {str(e)[:200]}"

    def _send(self, s):
        import torch.distributed as dist
        p0, p1, a, b = self.ranges[s]
        self.bank.gather_grads(p0, p1)                     # this segment's fresh gradients -> their windows (main stream)
        seg = self.bank.flat_grad[a:b]
        self.sent[s] = True
        self.this_order.append(s)
        if self.stream is None:
            self.works.append(dist.all_reduce(seg, async_op=True))
            return
        with torch.cuda.device(self.device):
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                dist.all_reduce(seg)

    def finish(self):
        """behind the backward pass: every segment is exchanged and visible to the main stream"""
        import torch.distributed as dist
        if self.suspended:
            dist.all_reduce(self.bank.flat_grad)            # one call (the gradients were gathered inside the graph / by the caller)
            return
        e0 = e1 = None
        if self.record_exposed and self.stream is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # what the hooks did not send, in the canonical order (the recorded order; first step: index order, descending)
        first = self.order is None
        canon = self.order if not first else list(range(len(self.ranges) - 1, -1, -1))
        for s in [s for s in canon if not self.sent[s]]:
            self._send(s)
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        if e0 is not None:
            e1.record()
            self.exposed_events.append((e0, e1))
        self.bank.point_grads_at_windows()
        bad = 1.0 if self.failed is not None else 0.0
        if first:
            # do the ranks agree on the order in which their backward passes complete the segments?  (segments no hook completes -
            # all parameters frozen - go last, in index order)
            seen = list(self.observed) + [s for s in canon if s not in self.observed]
            mine = torch.tensor(seen if self.failed is None else [-2] * len(self.ranges), dtype=torch.int32, device=self.device)
            every = [torch.empty_like(mine) for _ in range(self.world)]
            if self.world > 1:
                dist.all_gather(every, mine)
            else:
                every = [mine]
            live = [e for e in every if int(e[0]) != -2]        # -2: that rank already failed (its flag says so below)
            if live and all(torch.equal(e, live[0]) for e in live):
                self.order = [int(v) for v in live[0].tolist()]     # a rank that failed in this step follows the others' order
            elif self.failed is None:
                bad = 1.0
                self.failed = "ranks complete their segments in different orders: " + str([e.tolist() for e in every])
        self._flag.fill_(bad)
        dist.all_reduce(self._flag, op=dist.ReduceOp.MAX)
        # The flag is read TWO steps later (a pinned copy + event): reading it now would make the host wait for the GPU every step.
        # Every rank reads the same flag at the same step, so they switch forms together; in between a failed rank keeps completing
        # its steps in finish(), in the recorded order.
        if self.stream is None:
            verdicts = [float(self._flag)]
        else:
            if self._hosts is None:                        # four pinned words, reused in turn (a pinned allocation per step is not free)
                self._hosts = [torch.empty(1, dtype=torch.float32, pin_memory=True) for _ in range(4)]
            host = self._hosts[self.steps_done % 4]
            host.copy_(self._flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pending.append((host, ev))
            verdicts = []
            while len(self._pending) > 2:
                h, e = self._pending.pop(0)
                e.synchronize()
                verdicts.append(float(h))
        if any(v > 0 for v in verdicts):
            self.suspended = True
            self.fallback_reason = self.failed or "another rank reported a failed gradient hook / a different segment order"
        self.steps_done += 1

    def exposed_ms(self, per_step: bool = False):
        """average milliseconds per step the main stream waited for the exchange behind the backward pass (the first step, whose
        segments are all sent in finish(), is left out); per_step: the list"""
        if not self.exposed_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.exposed_events]
        if per_step:
            return ms
        ms = ms[1:] if len(ms) > 1 else ms
        return sum(ms) / len(ms)


def forward_backward(st: TrainingState, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """The bracket of a flat-mode step that a HIP graph can hold: 16-bit parameter copies, zeroed flat gradient array,
    autocast forward, loss, backward.  With `st.world` ranks the loss is scaled by 1 / world before the backward pass so that
    the SUM all-reduce in finish_step() leaves the mean gradient (what DDP's bucket division does)."""
    overlapped = st.exchange is not None and not st.exchange.suspended
    with st.bank.step():
        st.bank.release_grads()                                    # trainer.py:445 sets every p.grad to None
        if overlapped:
            st.exchange.arm()
        with torch.autocast(image.device.type, dtype=st.autocast_dtype,
                            enabled=image.device.type == "cuda" or st.scaler is not None):
            pred = st.model(image)
            loss = st.loss_fn(pred, label)                         # 3_train.py:62
        out = loss / st.world if st.world > 1 else loss
        if st.scaler is not None:
            out = out * st.scaler._scale                               # GradScaler.scale(loss): a device-side multiply
        out.backward()
        if not overlapped:
            st.bank.gather_grads()                                 # the fresh gradient tensors -> the flat array, one launch
    return loss.detach()


def finish_step(st: TrainingState) -> None:
    """exchange + update of a flat-mode step: all-reduce (world > 1), clip 12 + SGD-Nesterov over the flat arrays, poly LR"""
    if st.exchange is not None:
        st.exchange.finish()                                        # segments already in flight behind the backward pass
    elif st.world > 1:
        import torch.distributed as dist
        dist.all_reduce(st.bank.flat_grad)                          # 269.7 MB fp32 over xGMI, one collective
    st.optimizer.step()                                             # fp16: unscale + inf check + skip inside (train_ops.FusedClipSGD)
    if st.scaler is not None:
        sc = st.scaler                                              # what GradScaler.update() does, with the fused step's found_inf
        torch._amp_update_scale_(sc._scale, sc._growth_tracker, st.found_inf, sc.get_growth_factor(), sc.get_backoff_factor(),
                                 sc.get_growth_interval())
    st.scheduler.step()
    st.step += 1


class GraphedStep:
    """forward_backward() captured once as a HIP graph on static input buffers and replayed per step.

    A training step is ~2000 kernel launches; enqueueing them eagerly costs the host ~60 ms per step back to back
    (profiles/r02_host_time.log) - as much as the GPU needs to run them.  Everything inside the bracket has fixed shapes,
    fixed addresses (flat gradient array, bank copies, the caching allocator's private graph pool) and no host-side decision
    (the convolution routing is frozen after the warm-up steps), so one `hipGraphLaunch` replaces the launches.  The
    all-reduce, the optimizer (its learning rate is a kernel argument that changes every step) and the scheduler stay eager."""

    def __init__(self, st: TrainingState, image: torch.Tensor, label: torch.Tensor, warmup: int = 3):
        if not st.flat:
            raise RuntimeError("GraphedStep needs the flat-gradient training state (build_training_state(flat=True))")
        self.st = st
        self.image = image.clone()
        self.label = label.clone()
        if st.exchange is not None:
            st.exchange.suspended = True                            # no host decisions inside a captured bracket: one-call exchange
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                              # lazy initialisation and the routing tuner run here, eagerly
            for _ in range(max(1, warmup)):
                forward_backward(st, self.image, self.label)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = forward_backward(st, self.image, self.label)
        # the graph holds raw pointers into the parameter bank's gather map / pack buffer (allocated eagerly during the warm-up,
        # outside the graph's pool): keep them alive with the graph and stop the bank from re-creating them
        bank = getattr(st, "bank", None)
        if bank is not None:
            self._bank_buffers = (bank._dmap, bank._dbuf, bank.flat16z)
            bank.frozen = True
        st.graphed = self

    def __call__(self, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        self.image.copy_(image)
        self.label.copy_(label)
        self.graph.replay()
        finish_step(self.st)
        return self.loss.clone()                                   # the graph's output tensor is overwritten by the next replay


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
