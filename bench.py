"""Benchmark of SegMamba's hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Step = one training step of BASELINE config 2/3: SegMamba(4 -> 4, depths [2,2,2,2], widths [48,96,192,384]) on a
synthetic BraTS batch of 2 volumes of 128x128x128x4 per GPU, bf16 autocast, cross-entropy loss, backward, gradient
clip 12, SGD(lr 1e-2, momentum 0.99, nesterov, wd 3e-5) step - the loop body of the reference trainer
(light_training/trainer.py:445-470, 3_train.py:51-66).  The forward + backward of a step replay as one captured HIP graph
(`--no-graph` / SEGM_GRAPH=0: eager launches; `config.launch` says which ran).  N > 1: one process per GPU over RCCL, batch per
GPU fixed (weak scaling); the gradients of a step are ONE flat fp32 array that is all-reduced in a single call (default), or
SEGM_DDP=torch wraps the model in DistributedDataParallel as the reference does (trainer.py:353-357; find_unused_parameters
off, 64 MB buckets as gradient views).  `config.ddp` states the mode, the RCCL version, per-rank step times and the measured
all-reduce time.  `--gpus N` without a torchrun environment
re-executes itself through `python -m torch.distributed.run --nproc-per-node N` (as the reference's launch.py:89-108
does with torchrun); under torchrun WORLD_SIZE must equal --gpus.

Rank 0 prints ONE JSON line.  `value` = volumes / s over all GPUs.  `roofline` = the selective-scan forward at
SegMamba's largest stage (B=2, D=96, N=16, L=64^3, same dtype as the step): algorithmic bytes (SURVEY.md §8d:
e*B*L*(5D+2N)) / its measured duration (HIP events on the launch stream), against 8 TB/s HBM; the backward, the
three-directions-per-launch form the training step uses and `valu` (the same launch against the instruction-issue bound that
actually limits it) are reported next to it; `roofline_fp32` repeats it with fp32 I/O.  `config1` / `config4` = BASELINE's other
two measured configurations (one Mamba block at 2 x 64^3 x 384; scans of 2^21 and 2^24 steps).  `traffic` = HBM-side bytes of one forward launch from rocprofv3 PMC counters (FETCH_SIZE + WRITE_SIZE,
separate passes, calibrated on kernels with known byte counts), read from profiles/scan_traffic.json - the file
tools/gpu_pmc_traffic.sh regenerates together with the commit it was measured at (counters cannot be read from inside
the timed process); null when the file does not cover the dtype.  `cpu_baseline` = the CPU oracle timed on this box's
host cores on bounded samples: the C port (oracle/scan_ref.c, OpenMP) of the scan at the roofline shape, the PyTorch port
of the reference's selective_scan_ref (a Python loop over time, as the reference's CPU path is) and the whole network
on that path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# MIOpen's find step benchmarks every applicable solver the first time it sees a convolution; its "naive" reference
# solvers take ~0.4 s per call on these 3-D shapes (profiles/r01_bench_step_kernels.txt) and are never the winner.
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def scan_traffic(dtype_name):
    """HBM-side bytes per forward launch at the roofline shape from the committed PMC summary (tools/gpu_pmc_traffic.sh)."""
    path = os.path.join(ROOT, "profiles", "scan_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        e = rec.get(dtype_name)
        if e:
            return {"bytes": int(e["bytes"]), "measured_at": rec.get("commit"), "date": rec.get("date"),
                    "method": rec.get("method")}
    except (OSError, ValueError, KeyError):
        pass
    return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)   # defaults = the command the driver runs
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--size", type=int, default=128, help="edge of the cubic volume (128 = the BASELINE config)")
    p.add_argument("--batch", type=int, default=2, help="volumes per GPU")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the BASELINE config 1 / config 4 measurements")
    p.add_argument("--no-dropin", action="store_true", help="skip the drop-in (reference loop, fp16 + GradScaler, eager) and fp16 flat-mode steps")
    p.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a captured HIP graph")
    p.add_argument("--cpu-baseline-full", action="store_true", help="add the whole-network 64^3 / 32^3 fwd+bwd CPU legs (~80 s)")
    p.add_argument("--cpu-dry-run", action="store_true",
                   help="plumbing check without a GPU: gloo, CPU tensors, a tiny SegMamba on the CPU emulation of the kernels "
                        "(tests/emu); the printed value is NOT a measurement")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (one process per GPU; reference light_training/launch.py:89-108)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def time_gpu(fn, iters, warmup=3):
    """average milliseconds per call, HIP events on the current stream (the stream the kernels are launched on)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def algorithmic_bytes_scan(batch, dim, L, N, esize, G=1, backward=False):
    """standalone selective scan, real A, variable B / C, with z, D, delta_bias.
    fwd = e B L (5D + 2GN)  [read u, delta, z, B, C; write out, out_z];  bwd = B L (e (8D + 2GN) + 8GN)."""
    if backward:
        return batch * L * (esize * (8 * dim + 2 * G * N) + 8 * G * N)
    return esize * batch * L * (5 * dim + 2 * G * N)


def scan_roofline(dtype, device):
    from segmamba_amd import lib as L, ops_raw
    hip = L.get_lib()
    B, D, N, Lq = 2, 96, 16, 64 ** 3
    g = torch.Generator(device=device).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=device, generator=g).to(dtype)
    u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
    delta = (0.5 * torch.rand(B, Lq, D, device=device, generator=g)).to(dtype)
    A = -0.5 * torch.rand(D, N, device=device, generator=g)
    Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
    Dv = torch.randn(D, device=device, generator=g)
    db = 0.5 * torch.rand(D, device=device, generator=g)
    es = u.element_size()

    def fwd():
        return ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True)
    f = fwd()

    def bwd():
        return ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True,
                                chunk=f["chunk"])
    ms_f = time_gpu(fwd, 20)
    ms_b = time_gpu(bwd, 10)
    # what the training step launches: the three directions of a Mamba v3 layer (as stored / reversed / slice-interleaved, nslices
    # 64 at this stage) as ONE grid with a direction axis (segm_selective_scan_{fwd,bwd}_multi) on three sets of tensors
    orders = [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)]
    sets = [dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=Dv, z=z, delta_bias=db)] + [
        dict(u=rn(B, Lq, D), delta=(0.5 * torch.rand(B, Lq, D, device=device, generator=g)).to(dtype), A=A.clone(), B=rn(B, Lq, N),
             C=rn(B, Lq, N), D=Dv.clone(), z=rn(B, Lq, D), delta_bias=db.clone()) for _ in range(2)]
    fcalls = [dict(s_, delta_softplus=True, channel_last=True, time_order=o, nslices=ns, need_out=True, need_ckpt=True)
              for s_, (o, ns) in zip(sets, orders)]
    f3 = ops_raw.scan_fwd_multi(hip, fcalls)
    bcalls = [dict(s_, dout=dout, out=r["out"], ckpt=r["ckpt"], delta_softplus=True, channel_last=True, time_order=o, nslices=ns,
                   chunk=r["chunk"]) for s_, (o, ns), r in zip(sets, orders, f3)]
    ms_f3 = time_gpu(lambda: ops_raw.scan_fwd_multi(hip, fcalls), 10)
    ms_b3 = time_gpu(lambda: ops_raw.scan_bwd_multi(hip, bcalls), 5)
    del sets, fcalls, bcalls, f3
    bytes_f = algorithmic_bytes_scan(B, D, Lq, N, es)
    bytes_b = algorithmic_bytes_scan(B, D, Lq, N, es, backward=True)
    gf3, gb3 = 3 * bytes_f / ms_f3 * 1e-6, 3 * bytes_b / ms_b3 * 1e-6
    gf, gb = bytes_f / ms_f * 1e-6, bytes_b / ms_b * 1e-6
    tr = scan_traffic({torch.bfloat16: "bf16", torch.float32: "fp32", torch.float16: "fp16"}[dtype])
    # The kernels are bound by instruction issue, not by HBM.  Round 3 measured what an instruction costs a SIMD
    # in these kernels (profiles/r03_probe_valu3.log, r03_scan_occupancy.log, r03_scan_ablations.log): ~9 cycles for v_exp_f32 / v_log /
    # v_rcp, ~4.4 for every other vector instruction - packed or not - and ~2 for a scalar / LDS / memory instruction, at ANY number of
    # resident waves (time grows linearly from 3 waves per SIMD on; the round-2 model - 8.3 / 2.25 per lane-op / 2.7 - came from a probe
    # whose plain ops rotated over the four register banks and counted nothing but vector instructions).  Per wave-step the aggregate
    # pass issues 18 transcendental + 37 other vector + ~17 scalar / LDS / memory instructions, the apply pass 20 + 58 + ~40
    # (tools/isa_mix.py on the shipped binary): 359 + 515 cycles.  Measured with 6 waves per SIMD of work: 340 - 357 and 503 - 528.
    # Round 4: the counts of the shipped binary (tools/isa_mix.py; checkpoints every 8 steps now) and, next to the shipped stream,
    # the FLOOR of the two-pass algorithm - what no implementation of it can go below, so that `frac` cannot improve by recounting
    # (VERDICT r03): 32 v_exp (one per step, state and pass) + 4 transcendentals of softplus and the gate at 9 cycles; the packed
    # recurrence 3 (aggregate) + 4 (apply) instructions per state pair = 56, and ~10 others (one load + convert per row stream
    # and pass, the two stores, the output sum) at 4.4 cycles: 324 + 246 + 44 = 614 cycles per wave-step.
    shipped = {"aggregate": [18, 38, 15], "apply": [20, 55, 31]}      # [transcendental, other vector, scalar / LDS / memory / wait]
    cyc_needed = round(sum(t * 9.0 + v * 4.4 + o * 2.0 for t, v, o in shipped.values()))
    cyc_floor = round(36 * 9.0 + 56 * 4.4 + 10 * 4.4)
    elems_per_simd = B * D * Lq / 64 / 1024                    # wave-steps per SIMD (256 CUs x 4 SIMDs)
    cyc_taken = ms_f * 1e-3 * 2.1e9 / elems_per_simd
    cyc_taken3 = ms_f3 * 1e-3 * 2.1e9 / (3 * elems_per_simd)
    # Round 5: the floor expressed as an HBM fraction - what `frac` could reach if the launch issued nothing but
    # the floor stream: algorithmic bytes / (wave-steps per SIMD x floor cycles / 2.1 GHz) against 8 TB/s.  ~0.30 with 16-bit I/O,
    # ~0.60 with fp32 I/O: BASELINE.json's ">= 0.60 of the HBM roofline" is out of reach of ANY exact two-pass scan at N = 16 in
    # bf16 on this issue rate; it is within reach of the fp32-I/O launch only.
    ceiling = bytes_f / (elems_per_simd * cyc_floor / 2.1e9) * 1e-9 / HBM_PEAK_GBPS
    # the backward's own floor, built the same way (per wave-step = one time step of 64 channels x 16 states, both passes):
    #   transcendental: 32 v_exp (a = exp(delta A) in the reverse-aggregate and the main pass) + 6 (softplus, its derivative's sigmoid,
    #                   the gate's sigmoid - recomputed, not stored) = 38 x 9 cycles
    #   packed fp32 per state pair: aggregate 3; main: h recompute 1, dh 1, du 1, (h - b) 1, dh (h - b) 1, ddelta 2, dA 1, dB / dC
    #                   products 2 + their share of the d-tile reduce-scatter ~4 = 15  ->  8 x 18 = 144 x 4.4 cycles
    #   row streams: 7 loads + converts (u, delta, z, out, dout, B, C), 5 stores (du, ddelta, dz, dB, dC) ~ 14 x 4.4 cycles
    cyc_floor_b = round(38 * 9.0 + 144 * 4.4 + 14 * 4.4)
    cyc_taken_b = ms_b * 1e-3 * 2.1e9 / elems_per_simd
    cyc_taken_b3 = ms_b3 * 1e-3 * 2.1e9 / (3 * elems_per_simd)
    ceiling_b = bytes_b / (elems_per_simd * cyc_floor_b / 2.1e9) * 1e-9 / HBM_PEAK_GBPS
    return {
        "bound": "valu", "priced_against": "hbm", "kernel": "selective_scan_fwd (scan_fwd_agg + scan_carry + scan_fwd_apply)",
        "shape": {"B": B, "D": D, "N": N, "L": Lq, "layout": "channel-last", "chunk": f["chunk"]},
        "achieved": round(gf, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gf / HBM_PEAK_GBPS, 4),
        "ms": round(ms_f, 4), "algorithmic_bytes": bytes_f,
        "ceiling": {"frac": round(ceiling, 4), "what": "the instruction-issue floor of the two-pass algorithm (valu.floor) expressed as a "
                    "fraction of the 8 TB/s HBM roofline at this I/O width: the most `frac` can reach; the target >= 0.60 needs fp32 I/O",
                    "frac_of_ceiling": round(gf / HBM_PEAK_GBPS / ceiling, 4)},
        "traffic": tr["bytes"] if tr else None, "traffic_source": tr,
        "valu": {"bound": "instruction issue",
                 "floor": {"cycles_per_wave_step": cyc_floor, "frac": round(cyc_floor / cyc_taken, 4),
                           "frac_three_directions_per_launch": round(cyc_floor / cyc_taken3, 4),
                           "what": "minimal instruction stream of the two-pass algorithm: 32 v_exp + 4 softplus / gate transcendentals, "
                                   "56 packed recurrence instructions, ~10 load / convert / store / sum"},
                 "cycles_needed_per_wave_step": cyc_needed,
                 "cycles_taken_per_wave_step_at_2.1GHz": round(cyc_taken, 1), "frac": round(cyc_needed / cyc_taken, 4),
                 "three_directions_per_launch": {"cycles_taken_per_wave_step": round(cyc_taken3, 1), "frac": round(cyc_needed / cyc_taken3, 4)},
                 "instruction_costs_cycles": {"transcendental": 9.0, "other_vector": 4.4, "scalar_lds_memory": 2.0,
                                              "source": "profiles/r03_probe_valu3.log, r03_scan_occupancy.log, r03_scan_ablations.log"},
                 "instructions_per_wave_step": shipped,
                 "measured_with_6_waves_per_simd": {"aggregate": [340, 357], "apply": [503, 528]}},
        "note": "bound = what limits the kernel: instructions issued per SIMD (one v_exp_f32 per step and state in each of the two "
                "passes plus the recurrence; ~4.4 cycles per vector instruction whatever the occupancy).  achieved / peak / frac price "
                "it against the HBM roofline BASELINE.json's metric names (algorithmic bytes / time vs 8 TB/s); valu.frac prices the "
                "same launch against the issue bound of the shipped instruction stream",
        "backward": {"achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBPS, 4), "ms": round(ms_b, 4),
                     "algorithmic_bytes": bytes_b,
                     "floor": {"cycles_per_wave_step": cyc_floor_b, "frac": round(cyc_floor_b / cyc_taken_b, 4),
                               "frac_three_directions_per_launch": round(cyc_floor_b / cyc_taken_b3, 4),
                               "ceiling_frac_of_hbm": round(ceiling_b, 4),
                               "what": "38 transcendentals (32 v_exp over the two passes + softplus / gate), 144 packed fp32 (18 per state "
                                       "pair: 3 aggregate + 11 main + ~4 reduce-scatter share), ~14 row-stream instructions"},
                     "main_kernel": {"file": "csrc/scan_bwd_w8.hip", "instructions_per_step": 290,
                                     "per_state_pair_and_8_steps": {"vector": 152, "transcendental": 16, "lds": 13, "other": 23},
                                     "round_3": {"file": "scan_bwd_pair.hip (removed)", "instructions_per_step": 510, "ms": 0.915},
                                     "source": "tools/isa_mix.py on the shipped binary (profiles/r04_scan_bwd_isa.txt)"},
                     "deterministic": "dB / dC: per-d-tile fp32 slabs added in tile order (no atomics)"},
        "three_directions_per_launch": {
            "what": "the launch the training step issues: forward / reversed / slice-interleaved scans of one layer as one grid",
            "fwd_ms": round(ms_f3, 4), "fwd_achieved": round(gf3, 1), "fwd_frac": round(gf3 / HBM_PEAK_GBPS, 4),
            "bwd_ms": round(ms_b3, 4), "bwd_achieved": round(gb3, 1), "bwd_frac": round(gb3 / HBM_PEAK_GBPS, 4),
            "algorithmic_bytes": {"fwd": 3 * bytes_f, "bwd": 3 * bytes_b}},
    }


def config1_mamba_block(device):
    """BASELINE config 1: one Mamba(d_model=384, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=64) block, forward and
    forward + backward on randn(2, 64^3, 384) under bf16 autocast.  Per direction the scan moves
    e B L (5D + 2N) bytes forward and B L (e (8D + 2N) + 8N) backward (D = 768)."""
    from segmamba_amd.mamba_simple import Mamba
    torch.manual_seed(0)
    B, Lq, dm, N = 2, 64 ** 3, 384, 16
    m = Mamba(d_model=dm, d_state=N, d_conv=4, expand=2, bimamba_type="v3", nslices=64).to(device)
    x = torch.randn(B, Lq, dm, device=device, generator=torch.Generator(device=device).manual_seed(0))
    g = torch.randn(B, Lq, dm, device=device, generator=torch.Generator(device=device).manual_seed(1)).to(torch.bfloat16)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m(x)

    def fwd_bwd():
        for p in m.parameters():
            p.grad = None
        xx = x.detach().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xx)
        y.backward(g)
    ms_f = time_gpu(fwd, 5, warmup=2)
    ms_fb = time_gpu(fwd_bwd, 5, warmup=2)
    D = 2 * dm
    bf = 3 * algorithmic_bytes_scan(B, D, Lq, N, 2)
    bb = 3 * algorithmic_bytes_scan(B, D, Lq, N, 2, backward=True)
    return {"workload": "Mamba(d_model=384,d_state=16,v3,nslices=64) on (2, 262144, 384), bf16 autocast, three directions",
            "fwd_ms": round(ms_f, 3), "fwd_bwd_ms": round(ms_fb, 3),
            "scan_algorithmic_bytes": {"fwd": bf, "bwd": bb},
            "whole_block_vs_scan_bytes_GBps": {"fwd": round(bf / ms_f * 1e-6, 1), "fwd_bwd": round((bf + bb) / ms_fb * 1e-6, 1)},
            "note": "the block also runs in_proj / conv1d / x_proj / dt_proj / out_proj; GB/s = scan bytes over whole-block time"}


def config4_long_scan(device):
    """BASELINE config 4: the standalone scan forward and backward at B=1, D=96, N=16, L = 2^21 (what the stem yields for a
    256^3 volume) and L = 2^24 (the figure BASELINE.json quotes), bf16 and fp32 I/O, channel-last rows."""
    from segmamba_amd import lib as L, ops_raw
    hip = L.get_lib()
    out = {}
    for Lq in (1 << 21, 1 << 24):
        for dtype, name in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
            B, D, N = 1, 96, 16
            g = torch.Generator(device=device).manual_seed(0)
            rn = lambda *s: torch.randn(*s, device=device, generator=g).to(dtype)
            u, z, dout = rn(B, Lq, D), rn(B, Lq, D), rn(B, Lq, D)
            delta = (0.5 * torch.rand(B, Lq, D, device=device, generator=g)).to(dtype)
            A = -0.5 * torch.rand(D, N, device=device, generator=g)
            Bm, Cm = rn(B, Lq, N), rn(B, Lq, N)
            Dv = torch.randn(D, device=device, generator=g)
            db = 0.5 * torch.rand(D, device=device, generator=g)

            def fwd():
                return ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, need_out=True, need_ckpt=True)
            try:
                f = fwd()
            except RuntimeError as e:
                out[f"L{Lq}_{name}"] = {"error": str(e)[:200]}
                continue

            def bwd():
                return ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True,
                                        chunk=f["chunk"])
            it = 6 if Lq < (1 << 23) else 3
            ms_f, ms_b = time_gpu(fwd, it, warmup=1), time_gpu(bwd, it, warmup=1)
            es = u.element_size()
            bf, bb = algorithmic_bytes_scan(B, D, Lq, N, es), algorithmic_bytes_scan(B, D, Lq, N, es, backward=True)
            out[f"L{Lq}_{name}"] = {"fwd_ms": round(ms_f, 3), "bwd_ms": round(ms_b, 3),
                                    "fwd_GBps": round(bf / ms_f * 1e-6, 1), "bwd_GBps": round(bb / ms_b * 1e-6, 1),
                                    "fwd_frac": round(bf / ms_f * 1e-6 / HBM_PEAK_GBPS, 4),
                                    "bwd_frac": round(bb / ms_b * 1e-6 / HBM_PEAK_GBPS, 4), "chunk": f["chunk"]}
            del u, z, dout, delta, Bm, Cm, f
            torch.cuda.empty_cache()
    out["shape"] = {"B": 1, "D": 96, "N": 16, "layout": "channel-last", "peak_GBps": HBM_PEAK_GBPS}
    return out


def inference_rate(state, device, size):
    """Forward-only SegMamba on one 4 x size^3 volume (eval, no_grad, bf16 autocast): the setting of the only throughput the
    reference publishes (README table 5: 1.51 case/s at 128^3, hardware not stated - context, not a baseline)."""
    model = state.model.module if hasattr(state.model, "module") else state.model
    x = torch.rand(1, 4, size, size, size, device=device)
    was_training = model.training
    model.eval()

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(x)
    ms = time_gpu(run, 10, warmup=3)
    # memory of the forward alone: the training state (gradients, momenta, the captured graph's pool) stays resident in this process,
    # so the figure is the peak ABOVE what is held before the call, plus the parameters an inference process would hold
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated(device)
    torch.cuda.reset_peak_memory_stats(device)
    run()
    torch.cuda.synchronize()
    act = (torch.cuda.max_memory_allocated(device) - held) / 2 ** 20
    par = sum(p.numel() * p.element_size() for p in model.parameters()) / 2 ** 20
    model.train(was_training)
    return {"cases_per_s": round(1e3 / ms, 2), "ms_per_case": round(ms, 3), "input": [1, 4, size, size, size],
            "peak_mem_mb": {"activations_peak_mb": round(act, 1), "parameters_fp32_mb": round(par, 1),
                            "parameters_16bit_twin_mb": round(par / 2, 1), "standalone_total_mb": round(act + 1.5 * par, 1),
                            "what": "peak of one forward above what the process held before it + fp32 parameters + their 16-bit copies",
                            "reference_published_mb": 6279},
            "published_reference": {"cases_per_s": 1.51, "hardware": "not stated (reference README table 5)"}}


def gpu_state(local_rank=0):
    """Clock / power state of this rank's GPU from rocm-smi (before and after the timed loop: a box whose clock sags under the
    power cap explains a slower line - VERDICT r04 item 2).  Never raises: {} when rocm-smi is missing or its output changes."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    out = {}
    try:
        r = subprocess.run([exe, "-d", str(local_rank), "--showclocks", "--showpower", "--showtemp", "--showperflevel"],
                           capture_output=True, text=True, timeout=20)
        txt = r.stdout
        m = re.search(r"sclk clock level:\s*\w+:?\s*\((\d+)\s*Mhz\)", txt, re.I)
        if m:
            out["sclk_mhz"] = int(m.group(1))
        m = re.search(r"mclk clock level:\s*\w+:?\s*\((\d+)\s*Mhz\)", txt, re.I)
        if m:
            out["mclk_mhz"] = int(m.group(1))
        m = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", txt, re.I)
        if m:
            out["power_w"] = float(m.group(1))
        m = re.search(r"Temperature \(Sensor (?:junction|edge)\) \(C\):\s*([\d.]+)", txt, re.I)
        if m:
            out["temp_c"] = float(m.group(1))
        m = re.search(r"Performance Level:\s*(\w+)", txt, re.I)
        if m:
            out["perf_level"] = m.group(1)
    except Exception as e:                                  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {str(e)[:80]}"
    return out


def dropin_step(device, size, batch, steps, warmup):
    """The loop the reference's 3_train.py actually runs, on the drop-in modules and nothing else of this repository's harness
    (VERDICT r04 item 3): `from model_segmamba.segmamba import SegMamba`, torch.autocast("cuda") = fp16, GradScaler,
    torch.optim.SGD(lr 1e-2, wd 3e-5, momentum .99, nesterov), nn.CrossEntropyLoss, clip_grad_norm_(12) - eager launches, no
    parameter bank, no flat gradients, no fused optimizer, no graph (light_training/trainer.py:65-67, 445-466; 3_train.py:46-62).
    Then the same arithmetic on this repository's flat-gradient state in fp16 (amp="fp16": unscale + inf check inside the fused clip +
    SGD pass), eager and as a graph replay."""
    import torch.nn as nn
    from model_segmamba.segmamba import SegMamba
    from segmamba_amd.trainer import GraphedStep, SyntheticBraTS, build_training_state, train_step
    out = {}
    data = SyntheticBraTS(batch, size, device, seed=42)

    def timed(step_fn):
        for _ in range(warmup):
            step_fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step_fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, float(loss)

    torch.manual_seed(0)
    model = SegMamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384]).to(device)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    scaler = torch.amp.GradScaler("cuda")
    ce = nn.CrossEntropyLoss()

    def reference_body():
        image, label = data.next()
        for p in model.parameters():
            p.grad = None
        with torch.autocast("cuda", enabled=True):
            loss = ce(model(image), label)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 12)
        scaler.step(opt)
        scaler.update()
        return loss.detach()
    ms, loss = timed(reference_body)
    out["reference_loop"] = {"what": "model_segmamba.segmamba.SegMamba + torch.autocast('cuda') [fp16] + GradScaler + torch.optim.SGD + "
                                     "clip_grad_norm_(12) + nn.CrossEntropyLoss, eager, no bank / flat / fused optimizer / graph",
                             "ms_per_step": round(ms, 3), "volumes_per_s": round(batch / ms * 1e3, 3), "loss": round(loss, 5),
                             "loss_scale": scaler.get_scale()}
    del model, opt, scaler
    torch.cuda.empty_cache()
    for key, graph in (("fp16_flat_eager", False), ("fp16_flat_graph", True)):
        try:
            st = build_training_state(device, amp="fp16")
            if graph:
                GraphedStep(st, *data.next())
            ms, loss = timed(lambda: train_step(st, *data.next()))
            out[key] = {"ms_per_step": round(ms, 3), "volumes_per_s": round(batch / ms * 1e3, 3), "loss": round(loss, 5),
                        "flat_gradients": st.flat, "loss_scale": st.scaler.get_scale(), "skipped_last_step": bool(float(st.found_inf))
                        if st.found_inf is not None else None}
            del st
        except Exception as e:                              # noqa: BLE001 - a side measurement must not take the line with it
            out[key] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.empty_cache()
    return out


def launch_forms(state, data, device, steps=10, warmup=3):
    """N = 1 only: the same training step in the three forms a scaling curve mixes - the timed region of this
    line replays a captured graph, the N > 1 default launches eagerly with one gradient hook per parameter and the exchange in
    segments on a side stream.  Measured here on ONE rank (a 1-rank RCCL group: hooks, side stream, sliced all-reduce all run), so
    that an efficiency figure can be read like for like: eager_hooks_ms(1 GPU) against ms_per_step(N GPUs)."""
    import torch.distributed as dist
    from segmamba_amd.trainer import SegmentedExchange, train_step
    out = {}

    def timed():
        for _ in range(warmup):
            train_step(state, *data.next())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            train_step(state, *data.next())
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / steps * 1e3, 3)
    if state.graphed is not None:
        out["graph_ms"] = timed()
    state.graphed = None
    out["eager_ms"] = timed()
    try:
        if not dist.is_initialized():
            import datetime
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    timeout=datetime.timedelta(seconds=120))
        state.exchange = SegmentedExchange(state.bank, 1, int(os.environ.get("SEGM_DDP_SEGMENTS", "4")))
        state.exchange.record_exposed = True
        out["eager_hooks_ms"] = timed()
        out["eager_hooks_exposed_ms"] = round(state.exchange.exposed_ms(), 3)
        out["eager_hooks_exposed_ms_per_step"] = [round(v, 2) for v in state.exchange.exposed_ms(per_step=True)]
        out["segments"] = len(state.exchange.ranges)
        out["segment_order"] = state.exchange.order
        out["fallback"] = state.exchange.fallback_reason
        state.exchange.close()
        state.exchange = None
    except Exception as e:                                  # noqa: BLE001
        out["eager_hooks_ms"] = f"{type(e).__name__}: {str(e)[:160]}"
    out["note"] = "one rank; eager_hooks = hooks armed + 4 segments all-reduced on a side stream (what N > 1 runs by default)"
    return out


def cpu_baseline(full=False):
    """The reference's CPU path on this box's host cores, bounded samples:
      value            the C port of the scan (oracle/scan_ref.c, fp64 arithmetic, OpenMP) at the roofline shape, forward
      scan_fwd_bwd     the same, forward + backward
      torch_ref        the PyTorch port of selective_scan_ref (a Python loop over time - what the reference's own CPU path is),
                       forward and forward + backward at L = 2048 (cost is linear in L)
      segmamba         the whole network on that path: forward at 32^3; with --cpu-baseline-full also forward at 64^3
                       (BASELINE config 0) and forward + backward at 32^3 (another ~80 s of host time; one such run is kept
                       in profiles/)."""
    from oracle import ref_ops, scan_ref
    from oracle.segmamba_cpu import cpu_reference_segmamba
    out = {}
    B, D, N, Lq = 2, 96, 16, 64 ** 3
    g = torch.Generator().manual_seed(0)
    mk = lambda L_: dict(u=torch.randn(B, D, L_, generator=g), z=torch.randn(B, D, L_, generator=g),
                         delta=0.5 * torch.rand(B, D, L_, generator=g), Bm=torch.randn(B, N, L_, generator=g),
                         Cm=torch.randn(B, N, L_, generator=g), g=torch.randn(B, D, L_, generator=g))
    A = -0.5 * torch.rand(D, N, generator=g)
    Dv, db = torch.randn(D, generator=g), 0.5 * torch.rand(D, generator=g)
    c = mk(Lq)
    args = (c["u"], c["delta"], A, c["Bm"], c["Cm"], Dv, c["z"], db)
    scan_ref.scan_fwd(*[a[..., :1024].contiguous() if a.dim() == 3 else a for a in args], delta_softplus=True)   # load / warm
    t0 = time.time()
    scan_ref.scan_fwd(*args, delta_softplus=True)
    tf = time.time() - t0
    t0 = time.time()
    scan_ref.scan_bwd(*args, c["g"], delta_softplus=True)
    tb = time.time() - t0
    bf, bb = algorithmic_bytes_scan(B, D, Lq, N, 4), algorithmic_bytes_scan(B, D, Lq, N, 4, backward=True)
    out.update({"value": round(bf / tf * 1e-9, 3), "unit": "GB/s", "cores": scan_ref.threads(), "kind": "port",
                "seconds": round(tf, 2),
                "sample": f"HEADLINE VALUE = the C port of selective_scan_ref (oracle/scan_ref.c, OpenMP, fp64 arithmetic; NOT the reference's "
                          f"own path: that is `torch_ref` below, a Python loop over time, ~500 x slower), forward, fp32 I/O, B={B} D={D} "
                          f"N={N} L={Lq} (the whole roofline shape), same algorithmic-bytes formula",
                "scan_fwd_bwd": {"value": round((bf + bb) / (tf + tb) * 1e-9, 3), "unit": "GB/s", "seconds": round(tf + tb, 2)}})
    # the reference's own CPU path is a Python loop over time: L = 2048
    Ls = 2048
    c = mk(Ls)
    leaves = [c["u"].requires_grad_(), c["delta"].requires_grad_()]
    t0 = time.time()
    y = ref_ops.selective_scan_ref(leaves[0], leaves[1], A, c["Bm"], c["Cm"], Dv, z=c["z"], delta_bias=db, delta_softplus=True)
    t1 = time.time()
    y.backward(c["g"])
    t2 = time.time()
    out["torch_ref"] = {"sample": f"THE REFERENCE-PATH FIGURE: the pure-PyTorch selective_scan_ref the north star names (restated in "
                                  f"oracle/ref_ops.py - /root/reference does not exist on the GPU box), fp32, B={B} D={D} N={N} L={Ls}; "
                                  "cost is linear in L: extrapolated to the roofline shape below",
                        "cores": torch.get_num_threads(), "fwd_seconds": round(t1 - t0, 2), "fwd_bwd_seconds": round(t2 - t0, 2),
                        "fwd_GBps": round(algorithmic_bytes_scan(B, D, Ls, N, 4) / (t1 - t0) * 1e-9, 5),
                        "seconds_per_stage0_scan_extrapolated": round((t2 - t0) * Lq / Ls, 1)}
    torch.manual_seed(0)
    net = cpu_reference_segmamba(in_chans=4, out_chans=4, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384])
    x64, x32 = torch.rand(1, 4, 64, 64, 64), torch.rand(1, 4, 32, 32, 32)
    t0 = time.time()
    with torch.no_grad():
        net(x32)
    t32f = time.time() - t0
    out["segmamba"] = {"sample": "SegMamba(4->4,[2,2,2,2],[48,96,192,384]) on the pure-PyTorch reference path (oracle modules), fp32",
                       "fwd_32cube_seconds": round(t32f, 2), "volumes_per_s_fwd_32cube": round(1.0 / t32f, 4)}
    # BASELINE config 0 (0_inference.py on one 64^3 x 4 volume, selective_scan_ref path): in the default line since round 3
    # (VERDICT r02 weak #6); ~25 s on 8 cores, less on the GPU box's host
    t0 = time.time()
    with torch.no_grad():
        net(x64)
    t64 = time.time() - t0
    out["segmamba"].update({"fwd_64cube_seconds": round(t64, 2), "volumes_per_s_fwd_64cube": round(1.0 / t64, 4),
                            "config0": "SegMamba forward on rand(1, 4, 64, 64, 64), CPU reference path"})
    if full:
        t0 = time.time()
        torch.nn.functional.cross_entropy(net(x32), torch.randint(0, 4, (1, 32, 32, 32))).backward()
        t32 = time.time() - t0
        out["segmamba"].update({"fwd_bwd_32cube_seconds": round(t32, 2), "volumes_per_s_fwd_bwd_32cube": round(1.0 / t32, 4)})
    return out


_T0 = time.time()


def _stamp(what):
    """wall-clock stamps on stderr (how long each part of the line takes; the JSON line on stdout is untouched)"""
    print(f"[bench.py +{time.time() - _T0:6.1f} s] {what}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} processes; "
                 f"run `python bench.py --gpus {world}` or let bench.py launch itself (no torchrun)")
    distributed = world > 1 or os.environ.get("SEGM_FORCE_DDP") == "1"     # the latter: exercise the DDP path on one GPU
    dry = args.cpu_dry_run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dry:
        # plumbing check only (launch, rendezvous, DDP wrap, barrier / max-over-ranks timing, the JSON line): CPU tensors go
        # through the CPU emulation build of the kernel sources (test infrastructure, tests/emu)
        from segmamba_amd import lib as SL
        from tests import emu_util
        SL._lib = emu_util.emu_lib()
        SL.on_device = lambda t: True
        device = torch.device("cpu")
        sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    if distributed:
        import torch.distributed as dist
        import datetime
        # a collective that never completes ends the job with a message after 3 minutes instead of sitting until the driver's limit
        dist.init_process_group(backend="gloo" if dry else "nccl", init_method="env://", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("SEGM_PG_TIMEOUT_S", "180"))))

    from segmamba_amd.trainer import DDP_SETTINGS, SyntheticBraTS, build_training_state, train_step
    model = None
    if dry:
        from segmamba_amd.segmamba import SegMamba
        torch.manual_seed(0)
        model = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 16, 16, 32], hidden_size=32)
        args.size, args.batch = 32, 1
    state = build_training_state(device, distributed, local_rank, model=model)
    data = SyntheticBraTS(args.batch, args.size, device, seed=42 + rank)      # trainer.py:331 seeds 42 + rank

    def step():
        image, label = data.next()
        return train_step(state, image, label)

    # The forward + backward bracket as ONE captured HIP graph (trainer.GraphedStep); the batch is copied into the graph's
    # static input buffers every step, the all-reduce / optimizer / scheduler run eagerly behind the replay.  The capture
    # warm-up runs the bracket without an optimizer step, so it is not part of --warmup.
    # N > 1: eager launches by default - the gradient exchange then runs in segments on a side stream while the backward pass is
    # still producing gradients (trainer.SegmentedExchange); a captured bracket can only be followed by ONE exposed all-reduce.
    # SEGM_GRAPH_DDP=1 times the graph + one-call form instead; SEGM_BENCH_OTHER_FORM=1 measures it behind the timed region.
    graph_note = "off (--no-graph / SEGM_GRAPH=0)"
    graph_ddp = os.environ.get("SEGM_GRAPH_DDP", "0") == "1"
    if state.exchange is not None and not graph_ddp:
        graph_note = "off (N > 1: eager launches, gradient exchange in segments overlapped with the backward pass)"
        state.exchange.record_exposed = True
    elif state.flat and not dry and not args.no_graph and os.environ.get("SEGM_GRAPH", "1") == "1":
        from segmamba_amd.trainer import GraphedStep
        try:
            GraphedStep(state, *data.next())
            graph_note = "hipGraph replay of forward + backward"
        except Exception as e:                              # noqa: BLE001 - any capture failure: eager launches, reason reported
            state.graphed = None
            torch.cuda.synchronize()
            graph_note = f"capture failed, eager launches: {type(e).__name__}: {str(e)[:200]}"
    elif not state.flat:
        graph_note = "off (no flat-gradient state: torch DDP wrapper or CPU dry run)"

    for _ in range(args.warmup):
        step()
    if distributed:
        dist.barrier()
    sync()
    state_before = gpu_state(local_rank) if (rank == 0 and not dry) else {}
    state_during = {}
    sampler = None
    if rank == 0 and not dry:                              # one rocm-smi reading WHILE the timed steps run (the GPU idles down within
        import threading                                   # milliseconds of the final synchronize): a thread, the host loop only replays a graph

        def _sample():
            time.sleep(0.25)
            state_during.update(gpu_state(local_rank))
        sampler = threading.Thread(target=_sample, daemon=True)
        sampler.start()
    if not dry:
        torch.cuda.reset_peak_memory_stats(device)         # peak of the TIMED steps (the capture warm-up and --warmup are behind us)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if distributed:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    # the only figures the reference publishes beside its 1.51 case/s: training / inference memory, 17 976 / 6 279 MB (README.md:15-16)
    peak_mem = None if dry else {
        "peak_allocated_mb": round(torch.cuda.max_memory_allocated(device) / 2 ** 20, 1),
        "peak_reserved_mb": round(torch.cuda.max_memory_reserved(device) / 2 ** 20, 1),
        "allocated_between_steps_mb": round(torch.cuda.memory_allocated(device) / 2 ** 20, 1),
        "what": "torch caching allocator on rank 0 over the timed steps: parameters fp32 + 16-bit twin, flat gradients, momenta, kept "
                "activations of a 2 x 4 x 128^3 batch, scan checkpoints / workspaces.  Under a graph replay the step's tensors live in "
                "the graph's private pool, which peak_allocated does not see: there peak_reserved is the footprint; an eager run "
                "(--no-graph) reports the step's own peak in peak_allocated (README: both, with SEGM_RECOMPUTE 0 / 1)",
        "launch": "graph replay" if state.graphed is not None else "eager",
        "recompute": os.environ.get("SEGM_RECOMPUTE", "0"),
        "reference_published_mb": {"training": 17976, "inference": 6279, "source": "reference README.md:15-16 (Table 5), batch and GPU not stated"}}
    if sampler is not None:
        sampler.join(timeout=30)
    state_after = gpu_state(local_rank) if (rank == 0 and not dry) else {}
    rank_ms = [elapsed / args.steps * 1e3]
    allreduce_ms = exposed_ms = other_form = None
    overlapped = state.exchange is not None and not state.exchange.suspended      # the form the timed region ran
    if distributed and state.exchange is not None and not dry:
        exposed_ms = state.exchange.exposed_ms() if state.exchange.record_exposed else None
        state.exchange.record_exposed = False
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        elapsed = max(float(x.item()) for x in every)                # MAX over ranks
        if state.flat and not dry:                                   # the exchange step on its own: one all-reduce of the flat gradients
            try:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dist.all_reduce(state.bank.flat_grad)
                torch.cuda.synchronize()
                e0.record()
                dist.all_reduce(state.bank.flat_grad)
                e1.record()
                e1.synchronize()
                allreduce_ms = round(e0.elapsed_time(e1), 3)
            except Exception as e:                          # noqa: BLE001 - the line survives a failed side measurement
                allreduce_ms = f"{type(e).__name__}: {str(e)[:120]}"

    out = None
    if rank == 0:
        _stamp(f"timed loop done: {elapsed / args.steps * 1e3:.2f} ms per step")
        vols = world * args.batch * args.steps
        ddp = None
        if distributed:
            if state.flat and overlapped:
                ddp = {"mode": f"flat: the fp32 gradient array exchanged in {len(state.exchange.ranges)} segments on a side stream while the "
                               "backward pass runs, no wrapper", "segments": len(state.exchange.ranges)}
            elif state.flat:
                ddp = {"mode": "flat: one all-reduce of the flat fp32 gradient array per step, no wrapper"}
            else:
                ddp = dict(DDP_SETTINGS, mode="torch DistributedDataParallel")
            ddp["backend"] = "gloo (cpu dry run)" if dry else "nccl = RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
            ddp["rank_ms_per_step"] = {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)}
            ddp["rendezvous"] = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
            ddp["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "TORCH_NCCL", "SEGM_DDP", "SEGM_GRAPH"))}
            if state.exchange is not None:
                ddp["segment_order"] = state.exchange.order
                ddp["fallback"] = state.exchange.fallback_reason      # None: the overlapped form ran all the way
            if state.flat:
                ddp["gradient_bytes"] = int(state.bank.flat_grad.numel()) * 4
                ddp["allreduce_ms"] = allreduce_ms           # the whole array as one call, on its own
                ddp["allreduce_exposed_ms"] = None if exposed_ms is None else round(exposed_ms, 3)   # main stream waiting behind the backward pass
                ddp["other_form"] = other_form
        amp_name = {torch.bfloat16: "bf16", torch.float16: "fp16"}.get(state.autocast_dtype, str(state.autocast_dtype))
        out = {
            "metric": f"volumes/sec fwd+bwd+step, SegMamba {args.size}^3x4 (whole job; divide by n_gpus for per-GPU)",
            "value": round(vols / elapsed, 4), "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp32 (cpu dry run)" if dry else amp_name,
            "data": "synthetic" if not dry else "synthetic; CPU DRY RUN on emulated kernels with a tiny model - not a measurement",
            "config": {"workload": (f"SegMamba(4->4,[2,2,2,2],[48,96,192,384]) train step, {args.batch}x4x{args.size}^3 per GPU, "
                                    f"{amp_name} autocast, CE loss, clip 12, SGD nesterov") if not dry else
                                   "tiny SegMamba, 1x4x32^3 per process, plumbing only",
                       "volumes_per_gpu": args.batch, "volume": [args.size] * 3, "parallelism": f"dp{world}", "ddp": ddp,
                       "loss": round(float(loss), 5), "peak_mem_mb": peak_mem,
                       # host-side arrangements that do not change the arithmetic: the step's 16-bit parameter copies in one launch
                       # (param_bank.py), 128^3 volumes with a padded channel stride (ops_raw.volume_empty)
                       "param_bank": state.bank is not None, "volume_pad": os.environ.get("SEGM_VOLUME_PAD", "1") == "1",
                       "flat_gradients": state.flat, "launch": graph_note,
                       "command": "python bench.py --gpus %d --steps %d --warmup %d" % (world, args.steps, args.warmup),
                       "gpu_state": {"before_timed_loop": state_before, "during_timed_loop": state_during, "after_timed_loop": state_after,
                                     "source": "rocm-smi --showclocks --showpower --showtemp --showperflevel, rank 0's GPU"}},
        }
        if not args.no_roofline and not dry:
            out["inference"] = inference_rate(state, device, args.size)
            out["roofline"] = scan_roofline(torch.bfloat16, device)
            out["roofline_fp32"] = scan_roofline(torch.float32, device)
            _stamp("inference + roofline done")
            if not args.no_configs and world == 1:
                if state.flat and not distributed:
                    try:
                        out["config"]["launch_forms"] = launch_forms(state, data, device)
                    except Exception as e:                  # noqa: BLE001
                        out["config"]["launch_forms"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
                _stamp("launch_forms done")
                state = data = None                           # (released, not deleted: the closing block below looks at `state`)
                torch.cuda.empty_cache()
                if not args.no_dropin:
                    try:
                        out["dropin_step"] = dropin_step(device, args.size, args.batch, min(args.steps, 10), min(args.warmup, 3))
                    except Exception as e:                  # noqa: BLE001
                        out["dropin_step"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                    torch.cuda.empty_cache()
                    _stamp("dropin_step done")
                for key, fn in (("config1", config1_mamba_block), ("config4", config4_long_scan)):
                    try:
                        out[key] = fn(device)
                    except Exception as e:                  # noqa: BLE001 - the step's number must not be lost to a side measurement
                        out[key] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if not args.no_cpu_baseline and world == 1 and not dry:       # the host-core baseline is reported at N = 1 only
            _stamp("configs done")
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_full)
            _stamp("cpu_baseline done")
    def emit(line):
        # ONE JSON line, and the LAST line of stdout: the vendor libraries print diagnostics through C stdio ("GridwiseOp: ..." from the
        # solvers MIOpen tries on the drop-in path), fully buffered when stdout is a pipe - flush them out first, and send whatever a
        # library prints at teardown to /dev/null
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                   # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
        try:
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 1)
        except OSError:
            pass

    # N > 1 only, LAST, behind everything the line needs: the other launch form - captured bracket + one
    # all-reduce behind it - for three steps, so that the line carries both forms.  It cannot cost the measurement: every rank
    # attempts the capture on its own (no collective inside), ONE all-reduce (MIN) of a success flag decides for all of them whether
    # the extra steps run, and a watchdog thread ends the job cleanly - rank 0 prints the finished line first - if this block does
    # not come back within SEGM_BENCH_OTHER_FORM_TIMEOUT_S seconds (default 90).  SEGM_BENCH_OTHER_FORM=0 skips it.
    if distributed and not dry and state is not None and state.exchange is not None and state.graphed is None and not args.no_graph \
            and os.environ.get("SEGM_BENCH_OTHER_FORM", "1") == "1" and state.exchange.fallback_reason is None:
        import threading
        done = threading.Event()
        limit = float(os.environ.get("SEGM_BENCH_OTHER_FORM_TIMEOUT_S", "90"))

        def _watchdog():
            if done.wait(limit):
                return
            if rank == 0:
                out["config"]["ddp"]["other_form"] = {"form": "hipGraph replay + one all-reduce behind it",
                                                      "skipped": f"did not finish within {limit:.0f} s; the line was printed by the watchdog"}
                emit(out)
            os._exit(0)
        dist.barrier()                                       # rank 0 arrives behind its extras (inference, roofline): start the clocks together
        threading.Thread(target=_watchdog, daemon=True).start()
        other_form = {"form": "hipGraph replay + one all-reduce behind it"}
        err = None
        try:
            from segmamba_amd.trainer import GraphedStep
            GraphedStep(state, *data.next())
        except Exception as e:                              # noqa: BLE001
            err = f"{type(e).__name__}: {str(e)[:160]}"
            state.graphed = None
            torch.cuda.synchronize()
        okf = torch.tensor([0.0 if err else 1.0], device=device)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if float(okf.item()) >= 1.0:
            try:
                step()
                dist.barrier(); sync()
                t1 = time.perf_counter()
                for _ in range(3):
                    step()
                dist.barrier(); sync()
                other_form["ms_per_step"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
            except Exception as e:                          # noqa: BLE001
                other_form["error"] = f"{type(e).__name__}: {str(e)[:160]}"
        else:
            state.graphed = None
            other_form.update(skipped="the capture failed on at least one rank", this_rank=err)
        done.set()
        if rank == 0:
            out["config"]["ddp"]["other_form"] = other_form
    if rank == 0:
        emit(out)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    elif not dry:
        import torch.distributed as dist2
        if dist2.is_available() and dist2.is_initialized():     # the 1-rank group of launch_forms()
            dist2.destroy_process_group()


if __name__ == "__main__":
    main()
