"""Where does the bf16 library path leave the fp32 route?  Per-module relative error of the forward activations and of the
gradients w.r.t. the module outputs, SegMamba(4->4,[2,2,2,2],[48,96,192,384]) at 64^3 (batch 1), same weights:
    fp32 (no autocast)   vs   bf16 autocast   [vs fp16 autocast: 3 more mantissa bits - rounding shrinks 8x, a logic error does not]
Prints one line per leaf module in execution order: rel = ||a - ref|| / ||ref||."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_segmamba.segmamba import SegMamba

DEV = "cuda"
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
base = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384])
sd = {k: v.clone() for k, v in base.state_dict().items()}
g = torch.Generator().manual_seed(1)
x = torch.rand(1, 4, size, size, size, generator=g).to(DEV)
y = torch.randint(0, 4, (1, size, size, size), generator=g).to(DEV)


def run(dtype, sim=False):
    if sim:
        from tests.helpers import bf16_storage_simulation
        with bf16_storage_simulation():
            return run(torch.float32)
    m = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384])
    m.load_state_dict(sd)
    m = m.to(DEV)
    acts, grads, order = {}, {}, []

    def hook(name):
        def f(mod, inp, out):
            if not torch.is_tensor(out):
                return
            acts[name] = out.detach().float().cpu()
            order.append(name)
            if out.requires_grad:
                out.register_hook(lambda gr, name=name: grads.__setitem__(name, gr.detach().float().cpu()))
        return f
    for name, mod in m.named_modules():
        if name and (len(list(mod.children())) == 0 or name.endswith(".mamba") or name.count(".") == 0 or name.endswith("conv_block")
                     or name.startswith("vit.gscs.") and name.count(".") == 2 or name.startswith("vit.stages.") and name.count(".") == 3):
            mod.register_forward_hook(hook(name))
    scale = 65536.0 if dtype == torch.float16 else 1.0
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits.float(), y)
    (loss * scale).backward()
    pg = {k: p.grad.detach().float().cpu() / scale for k, p in m.named_parameters()}
    grads = {k: v / scale for k, v in grads.items()}
    return acts, grads, order, pg, float(loss)


ref = run(torch.float32)
print("loss fp32", ref[4])
res = {"bf16": run(torch.bfloat16), "sim": run(torch.float32, sim=True)}
if os.environ.get("WITH_FP16", "1") == "1":
    res["fp16"] = run(torch.float16)
for k, r in res.items():
    print("loss", k, r[4])


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


print("%-52s %10s | " % ("module (execution order)", "|act|") + " ".join("act_%s   dgrad_%s" % (k, k) for k in res))
for name in ref[2]:
    row = "%-52s %10.3e | " % (name, float(ref[0][name].norm()))
    for k, r in res.items():
        a = rel(r[0][name], ref[0][name]) if name in r[0] else float("nan")
        d = rel(r[1][name], ref[1][name]) if name in r[1] and name in ref[1] else float("nan")
        row += "%8.4f %8.4f   " % (a, d)
    print(row)
print("parameter gradients: rel err, worst 15 with |ref| > 1e-3 max")
gmax = max(float(v.norm()) for v in ref[3].values())
for k, r in res.items():
    rows = sorted(((rel(r[3][n], ref[3][n]), n) for n in ref[3] if float(ref[3][n].norm()) > 1e-3 * gmax), reverse=True)
    print(k, "median %.4f" % rows[len(rows) // 2][0], [(round(a, 3), n) for a, n in rows[:15]])
