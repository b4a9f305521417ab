"""Which host lines issue the step's small ATen launches: one eager training step under a TorchDispatchMode that records every
aten op whose tensors are all small (< 2^20 elements) with the innermost segmamba_amd frame of the Python stack (ops of C++
autograd nodes have none: 'autograd').  Prints the sites by launch count."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step

dev = torch.device("cuda", 0)
state = build_training_state(dev, False, 0)
data = SyntheticBraTS(2, 128, dev, seed=42)
for _ in range(2):
    train_step(state, *data.next())
torch.cuda.synchronize()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select",
        "aten.as_strided", "aten.expand", "aten.reshape", "aten.unsqueeze", "aten.squeeze", "aten.split", "aten.unbind", "aten.empty", "aten.new_empty",
        "aten.empty_like", "aten.empty_strided", "aten.is_", "aten.sym_", "aten.size", "aten.stride", "aten.unflatten", "aten.flatten", "aten.chunk",
        "aten.narrow", "aten.lift_fresh", "aten._local_scalar", "aten.item", "aten.record_stream", "aten.result_type", "aten.set_")
agg = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            for a in args:
                if isinstance(a, (list, tuple)):
                    ts += [t for t in a if isinstance(t, torch.Tensor)]
            if ts and all(t.numel() < (1 << 20) for t in ts) and any(t.is_cuda for t in ts):
                site = "autograd"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "segmamba_amd" in fr.filename:
                        site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                        break
                agg[(name, str([tuple(t.shape) for t in ts[:3]])[:60], site)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    train_step(state, *data.next())
torch.cuda.synchronize()
tot = sum(agg.values())
print("small ATen ops in one eager step:", tot)
by_site = collections.Counter()
for (name, shp, site), n in agg.items():
    by_site[site] += n
for site, n in by_site.most_common(40):
    print("%4d  %s" % (n, site))
print()
for (name, shp, site), n in agg.most_common(70):
    print("%4d  %-28s %-60s %s" % (n, name[:28], shp, site))
