"""One training step under torch.profiler: per-operator (with input shapes) device time, to attribute library kernels."""
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
for _ in range(2):
    train_step(state, *data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(state, *data.next())
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.self_device_time_total)[:70]
tot = sum(e.self_device_time_total for e in ka)
print(f"total self device time {tot / 1e3:.1f} ms")
for e in rows:
    print(f"{e.self_device_time_total / 1e3:8.2f} ms n={e.count:4d} {e.key[:50]:50s} {str(e.input_shapes)[:150]}")
