"""How host-bound is the training step?  Wall time per step against the time the Python side needs just to ENQUEUE a step
(returning from train_step without synchronising), and the enqueue time of the forward alone."""
import os, sys, time
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step
dev = torch.device("cuda", 0)
st = build_training_state(dev, False, 0)
data = SyntheticBraTS(2, 128, dev, seed=42)
for _ in range(3):
    train_step(st, *data.next())
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(st, *data.next())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("one step at a time (queue empty at the start): enqueue %.1f ms, until the GPU is done %.1f ms" % (sorted(enq)[len(enq) // 2], sorted(tot)[len(tot) // 2]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    train_step(st, *data.next())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("8 steps back to back: enqueue %.1f ms per step, wall %.1f ms per step" % ((t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3))
