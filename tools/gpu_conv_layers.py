"""Per-layer GPU timing of every 3x3x3 layer of the benchmarked network (B = 2, 128^3 input) through the product's routing
(segmamba_amd.conv3d: forward, data gradient, weight gradient), with the TF/s of each - where the convolution time of a step sits.
    python tools/gpu_conv_layers.py [out.txt]"""
import os, sys
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import conv3d as C

dev = "cuda"
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


# (name, Cin, Cout, S, count in the network)
LAYERS = [("enc1.conv2 / dec1 / dec2.conv2", 48, 48, 128, 4), ("dec2.conv1 (cat)", 96, 48, 128, 1),
          ("gsc0 x2 ", 48, 48, 64, 2), ("enc2.conv1", 48, 96, 64, 1), ("enc2.conv2 / dec3.conv2", 96, 96, 64, 2), ("dec3.conv1 (cat)", 192, 96, 64, 1),
          ("gsc1 x2", 96, 96, 32, 2), ("enc3.conv1", 96, 192, 32, 1), ("enc3.conv2 / dec4.conv2", 192, 192, 32, 2), ("dec4.conv1 (cat)", 384, 192, 32, 1),
          ("gsc2 x2", 192, 192, 16, 2), ("enc4.conv1", 192, 384, 16, 1), ("enc4.conv2 / dec5.conv2", 384, 384, 16, 2), ("dec5.conv1 (cat)", 768, 384, 16, 1),
          ("gsc3 x2", 384, 384, 8, 2), ("enc5.conv1", 384, 768, 8, 1), ("enc5.conv2", 768, 768, 8, 1)]
B = 2
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
by_size = {}
say("layer, Cin -> Cout @S^3 x count: forward / data gradient / weight gradient in ms (TF/s)")
for name, cin, cout, S, cnt in LAYERS:
    x = torch.randn(B, cin, S, S, S, device=dev).bfloat16()
    dy = torch.randn(B, cout, S, S, S, device=dev).bfloat16()
    w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device=dev)).bfloat16()
    fl = 2.0 * B * S ** 3 * cin * cout * 27

    def fwd():
        key, cands, variants = C._fwd_candidates(x, w, None, 1)
        return C._pick(key, cands, variants, x.shape[4])

    ms = [t(fwd), t(lambda: C._dgrad(dy, w, x, 1)), t(lambda: C._wgrad(x, dy, w, 1, torch.float32))]
    say("%-32s %3d -> %3d @%3d^3 x%d:  %7.3f (%4.0f)   %7.3f (%4.0f)   %7.3f (%4.0f)" % (
        name, cin, cout, S, cnt, ms[0], fl / ms[0] * 1e-9, ms[1], fl / ms[1] * 1e-9, ms[2], fl / ms[2] * 1e-9))
    for k, m in zip(("fwd", "dgrad", "wgrad"), ms):
        tot[k] += m * cnt
        by_size[S] = by_size.get(S, 0.0) + m * cnt
    del x, dy, w
say("per step: forward %.2f ms, data gradient %.2f ms, weight gradient %.2f ms (enc1.conv1 4 -> 48 and the first layer's data gradient not listed)" % (
    tot["fwd"], tot["dgrad"], tot["wgrad"]))
say("by volume width: " + ", ".join("%d^3 %.2f ms" % (s, v) for s, v in sorted(by_size.items(), reverse=True)))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
