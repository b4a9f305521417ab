"""The data-parallel path on the GPU: one process over RCCL (world size 1 - the box has one GPU), launched the way the driver
launches `bench.py --gpus N` (torch.distributed.run, 127.0.0.1 rendezvous).  What this covers that the gloo tests on CPU cannot:
the per-parameter hooks, the side stream and its events, `all_reduce` on slices of the flat gradient array through RCCL, the
barrier / max-over-ranks timing and the `config.ddp` block of the JSON line.  Reference: light_training/trainer.py:353-357."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_line(extra_env, port):
    env = dict(os.environ, SEGM_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--size", "64",
           "--no-cpu-baseline", "--no-roofline", "--no-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_segmented_exchange_over_rccl_world_size_one():
    d = _bench_line({}, 29531)
    ddp = d["config"]["ddp"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and ddp["mode"].startswith("flat")
    assert ddp["segments"] >= 2 and ddp["allreduce_exposed_ms"] is not None and ddp["allreduce_exposed_ms"] >= 0.0
    assert "eager" in d["config"]["launch"], d["config"]["launch"]
    assert ddp["allreduce_ms"] > 0


def test_graph_bracket_with_one_call_exchange_over_rccl_world_size_one():
    d = _bench_line({"SEGM_GRAPH_DDP": "1"}, 29532)
    assert "hipGraph" in d["config"]["launch"], d["config"]["launch"]
    assert d["value"] > 0 and d["config"]["ddp"]["allreduce_ms"] > 0
