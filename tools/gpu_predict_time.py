"""Prediction throughput on a BraTS-sized case: 128^3 sliding window (overlap 0.5, gaussian), with and without 8-way mirroring."""
import os, sys, time
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.segmamba import SegMamba
from segmamba_amd import predictor as P

dev = torch.device("cuda")
torch.manual_seed(0)
model = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384]).to(dev).eval()
x = torch.rand(1, 4, 138, 176, 144, device=dev)           # a typical BraTS foreground crop of 155 x 240 x 240
for swb in (2, 8):
    inferer = P.SlidingWindowInferer(roi_size=[128, 128, 128], sw_batch_size=swb, overlap=0.5, mode="gaussian")
    for axes in (None, [0, 1, 2]):
        pred = P.Predictor(window_infer=inferer, mirror_axes=axes)
        pred.maybe_mirror_and_predict(x, model, device=dev)          # warm-up (autotune, MIOpen find)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            y = pred.maybe_mirror_and_predict(x, model, device=dev)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"sw_batch {swb} mirror {axes}: {dt * 1e3:.1f} ms per case = {1 / dt:.2f} cases/s "
              f"({8 * (8 if axes else 1)} window forwards)", flush=True)
