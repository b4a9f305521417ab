"""TEST INFRASTRUCTURE: one SegMamba training step (smoke configuration, 1 x 4 x 32^3) with EVERY library route taken - the
kernels running on the CPU emulation of HIP (tests/emu) - against the same step on the ATen CPU path, in fp32 and in bf16
(the bf16 ATen run is the yardstick for what bf16 rounding alone does to the small gradients).

    python tools/emu_full_model_check.py [conv candidate index, default -1 = the last library candidate] [--linear]

`--linear` also switches the Mamba block's projections to segm_linear_rows (the SEGM_LINEAR_HIP route).

Slow (about ten minutes on 8 cores): an integration check for the build container, where there is no GPU; not part of the
pytest suite.  The Mamba operators have no CPU path, so all three runs use the (oracle-checked) emulated scan kernels."""
import os
import sys
import time

os.environ.setdefault("SEGM_CONV_FWD_UNTIMED", "1")     # every forward-convolution variant is a candidate here

sys.path.insert(0, ".")
import torch  # noqa: E402

from tests import emu_util  # noqa: E402
from segmamba_amd import lib as L, conv3d as C3, linear as LN, train_ops  # noqa: E402
from model_segmamba.segmamba import SegMamba  # noqa: E402

args = [a for a in sys.argv[1:] if a != "--linear"]
idx = int(args[0]) if args else -1
linear_route = "--linear" in sys.argv
torch.manual_seed(0)
cfg = dict(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 16, 16, 32], hidden_size=32)
ref = SegMamba(**cfg)
vol = torch.rand(1, 4, 32, 32, 32)
lab = torch.randint(0, 4, (1, 32, 32, 32))
L._lib = emu_util.emu_lib()


def run(bf16: bool, library: bool):
    m = SegMamba(**cfg)
    m.load_state_dict(ref.state_dict())
    if bf16:
        m = m.bfloat16()
    L.on_device = (lambda t: True) if library else (lambda t: False)
    LN._ROWS_HIP, LN._ROWS_MIN = (library and linear_route), 1
    C3._pick = lambda key, cands: cands[max(idx, -len(cands)) if library else 0]()
    t0 = time.time()
    out = m(vol.bfloat16() if bf16 else vol)
    loss = train_ops.cross_entropy(out, lab) if library else torch.nn.functional.cross_entropy(out.float(), lab)
    loss.backward()
    print(f"{'bf16' if bf16 else 'fp32'} {'library routes' if library else 'ATen routes   '}: loss {loss.item():.5f} ({time.time() - t0:.0f} s)", flush=True)
    return m, float(loss.detach()), {k: p.grad.float().clone() for k, p in m.named_parameters()}


_, l32, g32 = run(False, False)
_, l16, g16 = run(True, False)
m, llib, glib = run(True, True)
assert abs(llib - l32) <= 2 * abs(l16 - l32) + 1e-3, (llib, l16, l32)
gmax = max(float(v.abs().max()) for v in g32.values())
worst = (0.0, 0.0, "")
for k in g32:
    scale = max(float(g32[k].abs().max()), 1e-3 * gmax)
    dev_lib = float((glib[k] - g32[k]).abs().max()) / scale
    dev_aten = float((g16[k] - g32[k]).abs().max()) / scale
    assert torch.isfinite(glib[k]).all(), k
    assert dev_lib <= 3.0 * dev_aten + 0.1, (k, dev_lib, dev_aten)     # instance norms over 8 .. 64 elements amplify bf16 noise
    worst = max(worst, (dev_lib, dev_aten, k))
print(f"largest gradient deviation from fp32: library {worst[0]:.3f} (ATen bf16 at the same parameter: {worst[1]:.3f}) at {worst[2]}")

opt = train_ops.FusedClipSGD(m.float().parameters(), lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True, max_norm=12.0)
for p in m.parameters():
    p.grad = p.grad.float()
before = [p.detach().clone() for p in m.parameters()]
opt.step()
moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(m.parameters(), before))
print(f"fused clip + SGD step: clip coefficient {float(opt.last_clip[0]):.4f}, gradient norm {float(opt.last_clip[1]):.4f}, |dp| {moved:.4f}")
assert moved > 0 and torch.isfinite(opt.last_clip).all()
print("full model through every library route on the emulator: OK")
