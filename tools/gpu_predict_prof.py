"""torch.profiler view of one sliding-window prediction (no mirroring): device time vs wall time, top operators."""
import os, sys, time
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.segmamba import SegMamba
from segmamba_amd import predictor as P
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda")
torch.manual_seed(0)
model = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384]).to(dev).eval()
x = torch.rand(1, 4, 138, 176, 144, device=dev)
inferer = P.SlidingWindowInferer(roi_size=[128, 128, 128], sw_batch_size=2, overlap=0.5, mode="gaussian")
pred = P.Predictor(window_infer=inferer, mirror_axes=None)
for _ in range(2):
    pred.maybe_mirror_and_predict(x, model, device=dev)
torch.cuda.synchronize()
# plain forwards for comparison
xb = torch.rand(2, 4, 128, 128, 128, device=dev)
with torch.no_grad(), torch.autocast("cuda"):
    model(xb); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        model(xb)
    torch.cuda.synchronize(); print(f"plain forward batch 2: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms", flush=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    pred.maybe_mirror_and_predict(x, model, device=dev)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
ka = prof.key_averages()
dev_total = sum(e.self_device_time_total for e in ka)
print(f"wall {wall * 1e3:.1f} ms (profiled), device busy {dev_total / 1e3:.1f} ms")
for e in sorted(ka, key=lambda e: -e.self_device_time_total)[:14]:
    print(f"{e.self_device_time_total / 1e3:8.2f} ms dev  {e.self_cpu_time_total / 1e3:8.2f} ms cpu  n={e.count:5d}  {e.key[:70]}")
