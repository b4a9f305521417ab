"""North star N1, the last variant: delta = W_dt . x_dbl[:, :R] formed inside the two forward scan passes (scan_fwd(dt_x=, dt_weight=))
against the dt_proj launch (segm_linear_rows over the first column group of x_dbl) + scan, at the roofline shape (stage 0: 2 x 64^3
tokens, 96 channels, R = 3, x_dbl rows of 40 columns) as one and as three directions per launch; parity of the written delta."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmamba_amd import lib as L, ops_raw
from bench import time_gpu

hip = L.get_lib()
dev = torch.device("cuda")
B, D, N, Lq, R, P8 = 2, 96, 16, 64 ** 3, 3, 40
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)]
sets = []
for _ in orders:
    x_dbl = rn(B * Lq, P8)
    x_dbl[:, R:4] = 0
    w16 = (0.3 * torch.randn(D, R, device=dev, generator=g)).bfloat16()
    w8 = torch.zeros(D, 8, device=dev, dtype=torch.bfloat16)
    w8[:, :R] = w16
    v3 = x_dbl.view(B, Lq, P8)
    sets.append(dict(u=rn(B, Lq, D), z=rn(B, Lq, D), x_dbl=x_dbl, w8=w8, w32=w16.float().contiguous(), dt_x=v3[:, :, :R],
                     A=-0.5 * torch.rand(D, N, device=dev, generator=g), B=v3[:, :, 4:4 + N], C=v3[:, :, 4 + N:4 + 2 * N],
                     D=torch.randn(D, device=dev, generator=g), db=0.5 * torch.rand(D, device=dev, generator=g),
                     delta=torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)))


def dt_proj(n):
    return [ops_raw.linear_rows(hip, s["x_dbl"][:, :8], s["w8"]).view(B, Lq, D) for s in sets[:n]]


def scan_calls(fused, deltas, n):
    out = []
    for i, (s, (o, ns)) in enumerate(zip(sets[:n], orders)):
        c = dict(u=s["u"], delta=s["delta"] if fused else deltas[i], A=s["A"], B=s["B"], C=s["C"], D=s["D"], z=s["z"], delta_bias=s["db"],
                 delta_softplus=True, channel_last=True, time_order=o, nslices=ns, need_out=True, need_ckpt=True)
        if fused:
            c.update(dt_x=s["dt_x"], dt_weight=s["w32"])
        out.append(c)
    return out


deltas = dt_proj(3)
rf = ops_raw.scan_fwd_multi(hip, scan_calls(True, None, 3))
rp = ops_raw.scan_fwd_multi(hip, scan_calls(False, deltas, 3))
torch.cuda.synchronize()
for i, s in enumerate(sets):
    d = (s["delta"].float() - deltas[i].float()).abs()
    same = float((s["delta"] == deltas[i]).float().mean())
    rel = float(d.max() / deltas[i].float().abs().max())
    oz = float((rf[i]["out_z"].float() - rp[i]["out_z"].float()).abs().max() / rp[i]["out_z"].float().abs().max())
    print(f"direction {i}: delta written by the scan == dt_proj launch on {same * 100:.3f} % of elements, max |diff| / max {rel:.2e}; out_z max rel diff {oz:.2e}")
    r2 = ops_raw.scan_fwd(hip, s["u"], s["delta"].clone(), s["A"], s["B"], s["C"], s["D"], s["z"], s["db"], True, channel_last=True,
                          time_order=orders[i][0], nslices=orders[i][1], need_out=True, need_ckpt=True)
    print(f"   scan on the written delta == fused launch bit for bit: out_z {torch.equal(r2['out_z'], rf[i]['out_z'])}, ckpt {torch.equal(r2['ckpt'], rf[i]['ckpt'])}")

for n in (1, 3):
    t_dt = time_gpu(lambda: dt_proj(n), 20)
    t_plain = time_gpu(lambda: ops_raw.scan_fwd_multi(hip, scan_calls(False, deltas, n)), 10)
    t_fused = time_gpu(lambda: ops_raw.scan_fwd_multi(hip, scan_calls(True, None, n)), 10)
    print(f"{n} direction(s) per launch: dt_proj {t_dt * 1e3:6.1f} us + scan {t_plain * 1e3:7.1f} us = {(t_dt + t_plain) * 1e3:7.1f} us   "
          f"scan with dt_proj inside {t_fused * 1e3:7.1f} us  ({(t_fused - t_plain) * 1e3:+6.1f} us in the scan against {t_dt * 1e3:5.1f} us of launches saved)")
