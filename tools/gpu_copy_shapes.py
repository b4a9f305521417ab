"""One training step under torch.profiler: aten::copy_ / contiguous / add / sum / bmm calls by input shapes (device time, count)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    train_step(state, *data.next())
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
for op in ("aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::bmm", "aten::mm", "aten::fill_"):
    rows = sorted([e for e in ka if e.key == op and e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows) / 1e3
    print(f"{op}: {sum(e.count for e in rows)} calls, {tot:.2f} ms")
    for e in rows[:14]:
        print(f"    {e.self_device_time_total / 1e3:7.3f} ms n={e.count:4d} {str(e.input_shapes)[:150]}")
