"""North star, "causal depthwise conv1d fused into the same launch": the forward scan that forms u = SiLU(conv1d(x) + b) inside its two
passes (scan_fwd(conv_weight=...)) against the separate conv1d launch + scan, at the roofline shape and as three directions per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu

hip = L.get_lib()
dev = torch.device("cuda")
B, D, N, Lq = 2, 96, 16, 64 ** 3
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)]
sets = []
for _ in orders:
    sets.append(dict(x=rn(B, Lq, D), z=rn(B, Lq, D), delta=(0.5 * torch.rand(B, Lq, D, device=dev, generator=g)).bfloat16(),
                     A=-0.5 * torch.rand(D, N, device=dev, generator=g), B=rn(B, Lq, N), C=rn(B, Lq, N),
                     D=torch.randn(D, device=dev, generator=g), db=0.5 * torch.rand(D, device=dev, generator=g),
                     cw=0.5 * torch.randn(D, 4, device=dev, generator=g), cb=0.1 * torch.randn(D, device=dev, generator=g)))
conv_calls = [dict(x=s["x"], weight=s["cw"], bias=s["cb"], silu=True, channel_last=True, time_order=o, nslices=ns) for s, (o, ns) in zip(sets, orders)]
us = ops_raw.conv1d_fwd_multi(hip, conv_calls)


def scan_calls(fused):
    out = []
    for s, u, (o, ns) in zip(sets, us, orders):
        c = dict(u=s["x"] if fused else u, delta=s["delta"], A=s["A"], B=s["B"], C=s["C"], D=s["D"], z=s["z"], delta_bias=s["db"],
                 delta_softplus=True, channel_last=True, time_order=o, nslices=ns, need_out=True, need_ckpt=True)
        if fused:
            c.update(conv_weight=s["cw"], conv_bias=s["cb"])
        out.append(c)
    return out


t_conv1 = time_gpu(lambda: ops_raw.conv1d_fwd_multi(hip, conv_calls[:1]), 20)
t_conv3 = time_gpu(lambda: ops_raw.conv1d_fwd_multi(hip, conv_calls), 20)
for n in (1, 3):
    t_plain = time_gpu(lambda: ops_raw.scan_fwd_multi(hip, scan_calls(False)[:n]), 10)
    t_fused = time_gpu(lambda: ops_raw.scan_fwd_multi(hip, scan_calls(True)[:n]), 10)
    t_conv = t_conv1 if n == 1 else t_conv3
    print(f"{n} direction(s) per launch: conv1d {t_conv * 1e3:6.1f} us + scan {t_plain * 1e3:7.1f} us = {(t_conv + t_plain) * 1e3:7.1f} us   "
          f"scan with the conv inside {t_fused * 1e3:7.1f} us  (+{(t_fused - t_plain) * 1e3:5.1f} us against {t_conv * 1e3:5.1f} us for the launch; "
          f"the conv output is still needed in memory for x_proj)")
