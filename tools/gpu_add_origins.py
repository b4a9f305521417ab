"""Where the step's element-wise ATen launches come from: one training step under torch.profiler with Python stacks; prints every
aten::add / add_ / copy_ / mul / sum with its input shapes, device time and the innermost segmamba_amd frames (events without a
Python stack run on the autograd thread: gradient fan-in adds)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
state = build_training_state(dev, False, 0)
data = SyntheticBraTS(2, 128, dev, seed=42)
for _ in range(2):
    train_step(state, *data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_step(state, *data.next())
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.name not in ("aten::add", "aten::add_", "aten::copy_", "aten::mul", "aten::sum", "aten::cat", "aten::contiguous", "aten::clone", "aten::zero_", "aten::fill_"):
        continue
    t = e.device_time_total
    if t <= 0:
        continue
    frames = [f for f in (e.stack or []) if "segmamba_amd" in f or "autograd" in f][:3]
    key = (e.name, str(e.input_shapes)[:70], " <- ".join(f.split("/")[-1][:60] for f in frames) or "(no python stack: autograd thread)")
    agg[key][0] += t
    agg[key][1] += 1
for (name, shp, st), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{t / 1e3:7.3f} ms n={n:3d} {name:12s} {shp:70s} {st}")
