"""GPU probe (not product code): time the conv stem of SegMamba with the Mamba mixer stubbed out.

Answers: how far is MIOpen's 3D conv + ATen instance-norm from the 1.5 vol/s target, and
which kernels dominate?  Run on the GPU box through gpurun.
"""
import sys, time, types, json
import torch, torch.nn as nn

stub = types.ModuleType("segmamba_amd.mamba_simple")
class Mamba(nn.Module):
    def __init__(self, d_model, **kw):
        super().__init__(); self.w = nn.Parameter(torch.ones(d_model))
    def forward(self, x): return x * self.w
stub.Mamba = Mamba
sys.modules["segmamba_amd.mamba_simple"] = stub
sys.path.insert(0, ".")
from segmamba_amd.segmamba import SegMamba

def run(B, S, channels_last, iters=5, warm=2, prof=False):
    dev = "cuda"
    torch.manual_seed(0)
    m = SegMamba(in_chans=4, out_chans=4, depths=[2,2,2,2], feat_size=[48,96,192,384]).to(dev)
    if channels_last: m = m.to(memory_format=torch.channels_last_3d)
    x = torch.rand(B,4,S,S,S, device=dev)
    if channels_last: x = x.contiguous(memory_format=torch.channels_last_3d)
    y = torch.randint(0,4,(B,S,S,S), device=dev)
    ce = nn.CrossEntropyLoss()
    def step():
        for p in m.parameters(): p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = ce(m(x), y)
        loss.backward()
        return loss
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): step()
    torch.cuda.synchronize(); dt = (time.time()-t0)/iters
    res = dict(B=B, S=S, channels_last=channels_last, s_per_step=dt, vol_per_s=B/dt,
               max_mem_GB=torch.cuda.max_memory_allocated()/2**30)
    print(json.dumps(res), flush=True)
    if prof:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as p:
            step(); torch.cuda.synchronize()
        print(p.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=90), flush=True)
    return res

if __name__ == "__main__":
    print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
    torch.backends.cudnn.benchmark = True
    out = []
    out.append(run(1, 64, False, iters=3, warm=2))
    out.append(run(2, 128, False, iters=3, warm=2, prof=True))
    out.append(run(2, 128, True, iters=3, warm=2, prof=True))
    json.dump(out, open("gpurun_out/probe_stem.json","w"))
