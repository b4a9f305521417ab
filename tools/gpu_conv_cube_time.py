"""segm_conv3d_k3_cube_fwd on the 16^3 / 8^3 (and 32^3) layers of the benchmarked network: parity against fp32 ATen and time per
plan (column tiles per wave x splits) next to the product's present route.      python tools/gpu_conv_cube_time.py [out.txt]"""
import os, sys
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import conv3d as C, lib as L, ops_raw

hip = L.get_lib()
dev = "cuda"
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


LAYERS = [(192, 192, 16), (192, 384, 16), (384, 384, 16), (768, 384, 16), (384, 384, 8), (384, 768, 8), (768, 768, 8),
          (96, 96, 32), (96, 192, 32), (192, 192, 32), (384, 192, 32)]
if os.environ.get("CUBE_LAYERS"):
    LAYERS = [tuple(int(v) for v in s.split(",")) for s in os.environ["CUBE_LAYERS"].split(";")]
B = 2
for cin, cout, S in LAYERS:
    x = torch.randn(B, cin, S, S, S, device=dev).bfloat16()
    w = (0.05 * torch.randn(cout, cin, 3, 3, 3, device=dev)).bfloat16()
    bias = torch.randn(cout, device=dev)
    fl = 2.0 * B * S ** 3 * cin * cout * 27
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    img = ops_raw.conv3d_cube_weight_image(hip, w)

    def route():
        key, cands, variants = C._fwd_candidates(x, w, None, 1)
        return C._pick(key, cands, variants, x.shape[4])

    ms0 = t(route)
    out = "%3d -> %3d @%2d^3: present route %.3f ms (%4.0f TF/s) | cube" % (cin, cout, S, ms0, fl / ms0 * 1e-9)
    R = cin // 32
    best = None
    for nt in (4, 3, 2):
        if cout % (32 * nt):
            continue
        pnt, ps, _ = ops_raw.conv3d_cube_plan(hip, B, cin, cout, S, S, S, nt, 0)
        tried = [d for d in range(1, R + 1) if R % d == 0]
        for s in tried:
            y = ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias, nt=nt, splits=s)
            err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            ms = t(lambda: ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias, nt=nt, splits=s))
            out += "  nt%d s%d%s %.3f (%4.0f, err %.1e)" % (nt, s, "*" if s == ps else "", ms, fl / ms * 1e-9, err)
            if best is None or ms < best[0]:
                best = (ms, nt, s)
    pnt, ps, _ = ops_raw.conv3d_cube_plan(hip, B, cin, cout, S, S, S)
    out += "  | plan nt%d s%d, best nt%d s%d %.3f ms = %.1fx" % (pnt, ps, best[1], best[2], best[0], ms0 / best[0])
    say(out)
    del x, w, ref, img
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
