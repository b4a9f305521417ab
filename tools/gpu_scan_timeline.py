"""Per-wave timeline of the two forward scan passes ("find the 25 % that is not instructions").

    SEGM_LIB_OUT=build/variants/libsegm_timeline.so SEGM_EXTRA_HIPCC_FLAGS=-DSEGM_SCAN_TIMELINE python -m segmamba_amd.build
    python tools/gpu_scan_timeline.py build/variants/libsegm_timeline.so [out.txt]

The variant library stamps s_memtime at wave entry / loop head / end of the first sub-tile / loop exit / wave exit, the constant
100 MHz clock at entry and exit and the hardware id, into a side buffer (csrc/scan_fast.h, WaveTimeline).  This script runs the
roofline shape (stage 0: B = 2, D = 96, N = 16, L = 64^3, channel-last) one direction and three directions per launch, bf16 and
fp32, and prints per kernel: launch span, the shader clock, prologue / first sub-tile / steady-state / epilogue cycles per wave,
start skew and finish spread of the waves, and per-SIMD load (waves, busy time, finish time)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmamba_amd import lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
from segmamba_amd import ops_raw  # noqa: E402

out_path = sys.argv[2] if len(sys.argv) > 2 else None
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


def pct(v, qs=(0, 10, 50, 90, 100)):
    return "  ".join("p%d %.0f" % (q, np.percentile(v, q)) for q in qs)


def analyse(name, rec, nsub, steps_per_wave):
    """rec: (waves, 8) uint64 of one kernel slot; rows never written are all zero"""
    live = rec[:, 4] != 0
    r = rec[live].astype(np.int64)
    if len(r) == 0:
        say(name, ": no stamps")
        return
    t0, t1, t2, t3, t4, rt0, rt1, hw = (r[:, i] for i in range(8))
    span_us = (rt1.max() - rt0.min()) / 100.0
    dur_rt = (rt1 - rt0).astype(np.float64)
    ok = dur_rt > 0
    clk = float(np.median((t4 - t0)[ok] / dur_rt[ok])) * 100e6              # s_memtime ticks per second
    say("%s: %d waves, launch span %.1f us (first wave entry -> last wave exit, 100 MHz clock), s_memtime ticks at %.3f GHz" % (
        name, len(r), span_us, clk * 1e-9))
    tot = (t4 - t0).astype(np.float64)
    pro, first, steady, epi = (t1 - t0), (t2 - t1), (t3 - t2) / float(nsub - 1), (t4 - t3)
    say("   wave lifetime cycles      : mean %.0f   %s" % (tot.mean(), pct(tot)))
    say("   prologue (entry -> loop)  : mean %.0f   %s   = %.1f %% of the lifetime" % (pro.mean(), pct(pro), 100 * pro.sum() / tot.sum()))
    say("   first sub-tile (8 steps)  : mean %.0f   %s" % (first.mean(), pct(first)))
    say("   steady sub-tile (8 steps) : mean %.0f   %s   = %.1f cycles per step" % (steady.mean(), pct(steady), steady.mean() / 8))
    say("   epilogue (loop -> exit)   : mean %.0f   %s" % (epi.mean(), pct(epi)))
    start = (rt0 - rt0.min()) / 100.0
    end = (rt1 - rt0.min()) / 100.0
    say("   wave start, us after the first : %s" % pct(start))
    say("   wave end, us after the first   : %s" % pct(end))
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    uk, inv = np.unique(key, return_inverse=True)
    nw = np.bincount(inv)
    busy = np.bincount(inv, weights=(rt1 - rt0) / 100.0)
    fin = np.zeros(len(uk))
    np.maximum.at(fin, inv, end)
    beg = np.full(len(uk), 1e30)
    np.minimum.at(beg, inv, start)
    say("   SIMDs seen %d (of 1024): waves per SIMD min %d / mean %.2f / max %d; histogram %s" % (
        len(uk), nw.min(), nw.mean(), nw.max(), dict(zip(*np.unique(nw, return_counts=True)))))
    say("   per-SIMD finish time, us  : %s   (mean %.1f = %.1f %% of the span: the rest is the tail)" % (pct(fin), fin.mean(), 100 * fin.mean() / span_us))
    say("   per-SIMD first start, us  : %s" % pct(beg))
    # concurrency: resident waves per SIMD over time, sampled
    ts = np.linspace(0, span_us, 41)[1:-1]
    conc = [(np.sum((start <= t) & (end > t))) / float(len(uk)) for t in ts]
    say("   resident waves per SIMD at 2.5 %% steps of the span: %s" % " ".join("%.1f" % c for c in conc))
    # what the wave-steps cost: cycles per wave-step as the SIMD sees it = busy cycles per SIMD / wave-steps per SIMD
    ws_per_simd = nw * steps_per_wave
    cyc_simd = fin * 1e-6 * clk
    say("   cycles per wave-step per SIMD (finish time x clock / wave-steps on it): mean %.0f   %s" % (
        (cyc_simd / ws_per_simd).mean(), pct(cyc_simd / ws_per_simd)))
    # sub-tile time as a function of the number of co-resident waves when the wave started
    co = np.array([np.sum((key == k) & (start <= s) & (end > s)) for k, s in zip(key[:4000], start[:4000] + 1e-3)])
    for c in np.unique(co):
        m = co == c
        say("      waves that started with %d wave(s) resident on their SIMD: %5d   steady sub-tile %.0f cycles, lifetime %.0f" % (
            c, m.sum(), steady[:4000][m].mean(), tot[:4000][m].mean()))


def run(dtype, three, ckpt=True):
    hip = L.get_lib()
    dev = torch.device("cuda")
    B, D, N, Lq = 2, 96, 16, 64 ** 3
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(dtype)
    mk = lambda: dict(u=rn(B, Lq, D), delta=(0.5 * torch.rand(B, Lq, D, device=dev, generator=g)).to(dtype),
                      A=-0.5 * torch.rand(D, N, device=dev, generator=g), B=rn(B, Lq, N), C=rn(B, Lq, N),
                      D=torch.randn(D, device=dev, generator=g), z=rn(B, Lq, D), delta_bias=0.5 * torch.rand(D, device=dev, generator=g))
    orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)][: 3 if three else 1]
    calls = [dict(mk(), delta_softplus=True, channel_last=True, time_order=o, nslices=ns, need_out=True, need_ckpt=ckpt) for o, ns in orders]
    r = ops_raw.scan_fwd_multi(hip, calls)                 # warm-up, and the chunk length
    chunk = r[0]["chunk"]
    nsub = chunk // 8
    waves = len(orders) * (B * (D // 32) * (Lq // chunk) // 2 + 8)
    buf = torch.zeros(2 * waves * 8, dtype=torch.int64, device=dev)
    dbg = hip.dll.segm_debug_set_timeline
    dbg.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dbg.restype = ctypes.c_int
    for _ in range(3):
        ops_raw.scan_fwd_multi(hip, calls)
    torch.cuda.synchronize()
    assert dbg(buf.data_ptr(), waves) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops_raw.scan_fwd_multi(hip, calls)
    e1.record()
    torch.cuda.synchronize()
    assert dbg(0, 0) == 0
    rec = buf.cpu().numpy().view(np.uint64).reshape(2, waves, 8)
    say("")
    say("=== %s%s I/O, %d direction(s) per launch, chunk %d (%d sub-tiles of 8 steps per wave-item, 2 items per wave); whole forward "
        "(aggregate + carries + apply) %.3f ms by events ===" % ("" if ckpt else "NO CHECKPOINT STORES, ", str(dtype).split(".")[1], len(orders), chunk, nsub, e0.elapsed_time(e1)))
    analyse("scan_fwd_agg_fast  ", rec[0], nsub, chunk)
    analyse("scan_fwd_apply_fast", rec[1], nsub, chunk)


say("device:", torch.cuda.get_device_name(0))
only = os.environ.get("TIMELINE_ONLY")                     # e.g. "bf16-1": one configuration
for dt in (torch.bfloat16, torch.float32):
    for three in (False, True):
        if only and only != "%s-%d" % ("bf16" if dt == torch.bfloat16 else "fp32", 3 if three else 1):
            continue
        run(dt, three)
        if os.environ.get("TIMELINE_NOCKPT") == "1":
            run(dt, three, ckpt=False)
if out_path:
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
