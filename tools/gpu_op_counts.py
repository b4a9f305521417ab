"""One training step under torch.profiler: aten operators by launch count (which host-level ops the small kernels come from)."""
import os, sys
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
state = build_training_state(dev, False, 0)
data = SyntheticBraTS(2, 128, dev, seed=42)
for _ in range(3):
    train_step(state, *data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step(state, *data.next())
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = [e for e in ka if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print("aten ops that launch kernels, by count:")
for e in rows[:40]:
    print(f"  n={e.count:5d}  device {e.self_device_time_total / 1e3:7.2f} ms  cpu {e.self_cpu_time_total / 1e3:7.2f} ms  {e.key}")
