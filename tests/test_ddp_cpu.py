"""N > 1 path on CPU: two gloo processes run the training-step harness (segmamba_amd/trainer.py) the way bench.py and the
reference trainer do - DDP wrap (find_unused_parameters=True), gradient all-reduce, clip, SGD step - and must stay in
lock-step with each other and with a single process seeing the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _tiny_model():
    torch.manual_seed(0)
    # stand-in with the same kinds of layers as the conv stem (the HIP kernels cannot run in a CPU process)
    return nn.Sequential(nn.Conv3d(4, 8, 3, padding=1), nn.InstanceNorm3d(8), nn.LeakyReLU(0.01), nn.Conv3d(8, 4, 1))


def _batch(rank):
    g = torch.Generator().manual_seed(42 + rank)
    return torch.rand(2, 4, 8, 8, 8, generator=g), torch.randint(0, 4, (2, 8, 8, 8), generator=g)


def _use_emulated_library():
    """route CPU tensors through the library's kernels on the CPU emulation of HIP (test infrastructure, tests/emu)"""
    from segmamba_amd import lib as L
    from tests import emu_util
    L._lib = emu_util.emu_lib()
    L.on_device = lambda t: True


def _net(st):
    return st.model.module if hasattr(st.model, "module") else st.model


def _worker(rank, world, port, out, library=False, ddp="torch", segments=None, suspend=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if library:
        _use_emulated_library()
    from segmamba_amd.trainer import build_training_state, train_step
    st = build_training_state(torch.device("cpu"), distributed=True, model=_tiny_model(), ddp=ddp, ddp_segments=segments)
    assert st.flat == (library and ddp == "flat") and st.world == (world if st.flat else 1)
    if st.flat and segments is not None:
        assert (st.exchange is not None) == (segments > 1)
        if st.exchange is not None:
            assert len(st.exchange.ranges) == min(segments, 4)       # four parameters: at most four segments
            st.exchange.suspended = suspend                          # what GraphedStep sets: hooks idle, one call in finish()
    img, lab = _batch(rank)
    for _ in range(2):
        loss = train_step(st, img, lab)
    # the bench's timing reduction: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out[rank] = (float(loss), [p.detach().clone() for p in _net(st).parameters()], float(t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_ddp_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][2] == out[1][2] == 2.0
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b), "ranks diverged"
    # single process, both ranks' batches concatenated: DDP averages the per-rank mean losses == mean over the union
    from segmamba_amd.trainer import build_training_state, train_step
    st = build_training_state(torch.device("cpu"), distributed=False, model=_tiny_model())
    img = torch.cat([_batch(0)[0], _batch(1)[0]])
    lab = torch.cat([_batch(0)[1], _batch(1)[1]])
    for _ in range(2):
        train_step(st, img, lab)
    for a, b in zip(out[0][1], st.model.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("ddp", ["torch", "flat"])
def test_two_process_ddp_with_library_loss_and_optimizer(monkeypatch, ddp):
    """The same lock-step check with the library's cross entropy and clip + SGD kernels (emulated).  ddp="torch": the reference's
    wrapper - gradients are views into DDP's buckets (gradient_as_bucket_view), not necessarily 16-byte aligned, the multi-tensor
    kernels read them where they are.  ddp="flat" (the GPU default): no wrapper - every rank scales its loss by 1 / world, ONE
    all-reduce sums the flat gradient array, the optimizer steps over the three flat arrays."""
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()                                   # before the workers race to build it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, True, ddp), nprocs=2, join=True)
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b), "ranks diverged"
    from segmamba_amd import lib as L
    from segmamba_amd.trainer import build_training_state, train_step
    from segmamba_amd.train_ops import FusedClipSGD
    st = build_training_state(torch.device("cpu"), distributed=False, model=_tiny_model())      # ATen loss / SGD, one process
    assert not isinstance(st.optimizer, FusedClipSGD)
    img = torch.cat([_batch(0)[0], _batch(1)[0]])
    lab = torch.cat([_batch(0)[1], _batch(1)[1]])
    for _ in range(2):
        train_step(st, img, lab)
    for a, b in zip(out[0][1], st.model.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    monkeypatch.setattr(L, "_lib", emu_util.emu_lib())
    monkeypatch.setattr(L, "on_device", lambda t: True)
    assert isinstance(build_training_state(torch.device("cpu"), model=_tiny_model()).optimizer, FusedClipSGD)


def test_segmented_exchange_equals_the_one_call_exchange_bit_for_bit():
    """ddp="flat": the gradient all-reduce in K segments started from post-accumulate hooks while the backward pass runs
    (trainer.SegmentedExchange; the reference's DDP overlaps its buckets the same way, light_training/trainer.py:353-357) must
    leave exactly the parameters of the one-call exchange - sums are element-wise - for K = 2, 3, 8 (more segments than
    parameters), and so must the suspended form a captured HIP graph uses (hooks idle, one call behind the bracket)."""
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()
    results = {}
    for key, (segments, suspend) in {"one": (1, False), "k2": (2, False), "k3": (3, False), "k8": (8, False), "graph": (3, True)}.items():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out, True, "flat", segments, suspend), nprocs=2, join=True)
        for a, b in zip(out[0][1], out[1][1]):
            assert torch.equal(a, b), f"ranks diverged ({key})"
        results[key] = (out[0][0], [t.clone() for t in out[0][1]])
    for key in ("k2", "k3", "k8", "graph"):
        assert results[key][0] == results["one"][0], key
        for a, b in zip(results[key][1], results["one"][1]):
            assert torch.equal(a, b), key


def _segmamba_worker(rank, world, port, out, ddp="torch"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _use_emulated_library()
    from segmamba_amd.segmamba import SegMamba
    from segmamba_amd.trainer import build_training_state, train_step
    torch.manual_seed(rank if ddp == "flat" else 0)             # flat: different initial weights per rank - rank 0's must win
    net = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 16, 16, 32], hidden_size=32)
    st = build_training_state(torch.device("cpu"), distributed=True, model=net, ddp=ddp)
    g = torch.Generator().manual_seed(42 + rank)
    img, lab = torch.rand(1, 4, 32, 32, 32, generator=g), torch.randint(0, 4, (1, 32, 32, 32), generator=g)
    losses = [float(train_step(st, img, lab)) for _ in range(2)]
    out[rank] = (losses, [p.detach().clone() for p in _net(st).parameters()],
                 [None if p.grad is None else bool(torch.isfinite(p.grad).all()) for p in _net(st).parameters()])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ddp", ["torch", "flat"])
def test_two_process_ddp_with_the_real_segmamba_on_emulated_kernels(ddp):
    """(ddp="flat": the same through the wrapper-free form - flat gradient array, one all-reduce, flat optimizer step.)
    The REAL network under DDP: SegMamba with its custom autograd Functions (MambaInnerCore, the conv dispatcher, fused norms)
    on the library's kernels (CPU emulation), gradients as views into DDP's buckets (gradient_as_bucket_view), the library's
    loss and clip + SGD kernels.  Every parameter must receive a gradient on both ranks (find_unused_parameters=False is only
    legal then) and the ranks must stay bit-identical after two steps."""
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_segmamba_worker, args=(2, port, out, ddp), nprocs=2, join=True)
    assert all(f is True for f in out[0][2]) and all(f is True for f in out[1][2]), "a parameter got no (finite) gradient"
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b), "ranks diverged"
    assert out[0][0] != out[1][0]                           # different data per rank (seed 42 + rank), same weights


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun) must start two ranks and report n_gpus = 2; under a launcher whose world size
    differs from --gpus it must refuse.  CPU dry run: gloo, emulated kernels, tiny model."""
    import json
    import subprocess
    import sys
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]                 # rank 0 prints ONE line
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["ddp"]["mode"].startswith("flat")
    assert rec["config"]["ddp"]["rank_ms_per_step"]["max"] >= rec["config"]["ddp"]["rank_ms_per_step"]["min"] > 0
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-dry-run", "--steps", "1", "--warmup", "0"],
                        capture_output=True, text=True, timeout=600, env=env2, cwd=root)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in r2.stderr


def test_bench_eight_ranks_cpu_dry_run():
    """the driver's scaling command shape - `bench.py --gpus 8` - as 8 gloo ranks on the emulated kernels:
    rendezvous on 127.0.0.1, the record-only first step, hook sends from the second, rank-0-only extras while seven ranks sit in
    the final barrier, ONE JSON line with n_gpus = 8 and no fallback."""
    import json
    import subprocess
    import sys
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--cpu-dry-run", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    rec = json.loads(line[0])
    ddp = rec["config"]["ddp"]
    assert rec["n_gpus"] == 8 and rec["config"]["parallelism"] == "dp8" and rec["scaling"] == "weak"
    assert ddp["mode"].startswith("flat") and ddp["fallback"] is None and sorted(ddp["segment_order"]) == list(range(ddp["segments"]))
    assert ddp["rendezvous"]["MASTER_ADDR"] == "127.0.0.1" and ddp["rendezvous"]["WORLD_SIZE"] == "8"
    assert rec["value"] > 0 and abs(rec["value"] - 8 * 1 / (rec["ms_per_step"] * 1e-3)) <= 1e-2 * rec["value"]


def _failing_worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    _use_emulated_library()
    from segmamba_amd.trainer import build_training_state, train_step
    st = build_training_state(torch.device("cpu"), distributed=True, model=_tiny_model(), ddp="flat", ddp_segments=3 if mode != "one" else 1)
    img, lab = _batch(rank)
    if mode == "raise" and rank == 1:
        ex, real, calls = st.exchange, st.exchange._send, []

        def flaky(s):
            calls.append(s)
            if len(calls) == 5:                              # the second segment of the second step, from inside the autograd hook
                raise RuntimeError("injected: slice not aligned")
            return real(s)
        ex._send = flaky
    if mode == "order" and rank == 1:
        ex, real = st.exchange, st.exchange._hook
        held = []

        def late(p):                                        # rank 1 completes its LAST-produced segment first: a different send order
            if ex.seg_of[id(p)] == len(ex.ranges) - 1 and ex.steps_done == 0:
                held.append(p)
                return
            real(p)
        for h in ex._hooks:
            h.remove()
        ex._hooks = [p.register_post_accumulate_grad_hook(late) for p in ex.bank.params if p.requires_grad]
    for _ in range(5):
        loss = train_step(st, img, lab)
    ex = st.exchange
    out[rank] = (float(loss), [p.detach().clone() for p in st.model.parameters()],
                 None if ex is None else (ex.suspended, ex.fallback_reason, ex.failed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["raise", "order"])
def test_segmented_exchange_falls_back_to_one_call_when_a_rank_fails(mode):
    """round 5: an exception inside rank 1's gradient hook - or ranks that complete their segments in different
    orders - must neither hang the other rank in a collective nor change the result: the failing step is completed in finish() in
    the recorded order, a flag all-reduce tells every rank, all of them go on in the one-call form, and after five steps the
    parameters are those of the one-call exchange, bit for bit, on both ranks."""
    from tests import emu_util
    if not emu_util.emu_available():
        pytest.skip("no host clang for the emulation build")
    emu_util.build_emu()
    res = {}
    for m in ("one", mode):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_failing_worker, args=(2, port, out, m), nprocs=2, join=True)
        res[m] = {r: out[r] for r in (0, 1)}
    for r in (0, 1):
        assert res[mode][r][0] == res["one"][r][0]
        for a, b in zip(res[mode][r][1], res["one"][0][1]):
            assert torch.equal(a, b), (mode, r)
        suspended, reason, failed = res[mode][r][2]
        assert suspended and reason, (mode, r, reason)
    assert res[mode][1][2][2] is not None                       # rank 1 knows what happened to it ...
    if mode == "raise":
        assert "injected" in res[mode][1][2][2] and res[mode][0][2][2] is None and "another rank" in res[mode][0][2][1]
