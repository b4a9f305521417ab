"""Benchmark of SegMamba's hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Step = one training step of BASELINE config 2/3: SegMamba(4 -> 4, depths [2,2,2,2], widths [48,96,192,384]) on a
synthetic BraTS batch of 2 volumes of 128x128x128x4 per GPU, bf16 autocast, cross-entropy loss, backward, gradient
clip 12, SGD(lr 1e-2, momentum 0.99, nesterov, wd 3e-5) step - the loop body of the reference trainer
(light_training/trainer.py:445-470, 3_train.py:51-66).  N > 1: DistributedDataParallel over RCCL
(find_unused_parameters=True as trainer.py:354-357), batch per GPU fixed (weak scaling).

Rank 0 prints ONE JSON line.  `value` = volumes / s over all GPUs.  `roofline` = the selective-scan forward at
SegMamba's largest stage (B=2, D=96, N=16, L=64^3, same dtype as the step): algorithmic bytes (SURVEY.md §8d:
e*B*L*(5D+2N)) / its measured duration (HIP events on the launch stream), against 8 TB/s HBM; the backward is reported
next to it.  `traffic` = HBM-side bytes of one forward launch from rocprofv3 PMC counters (FETCH_SIZE + WRITE_SIZE,
separate passes, calibrated on kernels with known byte counts: profiles/r01_scan_pmc_hbm_traffic.txt) - a constant
measured once per shape, since counters cannot be read from inside the timed process.  It is ~2x the algorithmic
bytes by construction: the chunked scan reads u / delta / B twice (aggregate + apply passes) and, in training mode,
writes the fp32 state checkpoints the backward starts from (as many bytes as out + out_z).  `cpu_baseline` = the CPU oracle (a port of the reference's pure-PyTorch selective_scan_ref) timed on this
box's host cores on a bounded sample of the same operator.
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
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per forward launch at the roofline shape, bf16 (tools/gpu_pmc_scan.sh ->
# profiles/r01_scan_pmc_hbm_traffic.txt; reads scaled by the 2-byte-row calibration 100.66 MB / 67.47 MB):
#   agg 167.3 MB*1.49 + 12.8 MB, carry 11.8 + 12.0 MB, apply 276.7 MB*1.49 + 394.1 MB
SCAN_FWD_TRAFFIC_BF16 = int((167.27 * 1.49 + 12.75 + 11.76 + 12.0 + 276.67 * 1.49 + 394.08) * 2 ** 20)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size", type=int, default=128, help="edge of the cubic volume (128 = the BASELINE config)")
    p.add_argument("--batch", type=int, default=2, help="volumes per GPU")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


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


def scan_roofline(dtype, device):
    from segmamba_amd import lib as L, ops_raw
    from oracle.ref_ops import algorithmic_bytes_scan
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
    bytes_f = algorithmic_bytes_scan(B, D, Lq, N, es)
    bytes_b = algorithmic_bytes_scan(B, D, Lq, N, es, backward=True)
    gf, gb = bytes_f / ms_f * 1e-6, bytes_b / ms_b * 1e-6
    return {
        "bound": "hbm", "kernel": "selective_scan_fwd (scan_fwd_agg + scan_carry + scan_fwd_apply)",
        "shape": {"B": B, "D": D, "N": N, "L": Lq, "layout": "channel-last", "chunk": f["chunk"]},
        "achieved": round(gf, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gf / HBM_PEAK_GBPS, 4),
        "ms": round(ms_f, 4), "algorithmic_bytes": bytes_f,
        "traffic": SCAN_FWD_TRAFFIC_BF16 if dtype == torch.bfloat16 else None,
        "note": "VALU-issue bound (one v_exp_f32 per step and state per pass, two passes): DESIGN.md section 4",
        "backward": {"achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBPS, 4), "ms": round(ms_b, 4),
                     "algorithmic_bytes": bytes_b},
    }


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
    model.train(was_training)
    return {"cases_per_s": round(1e3 / ms, 2), "ms_per_case": round(ms, 3), "input": [1, 4, size, size, size],
            "published_reference": {"cases_per_s": 1.51, "hardware": "not stated (reference README table 5)"}}


def cpu_baseline():
    """The oracle's selective_scan_ref (fp32, pure PyTorch - a port of the reference's CPU path) on the host cores."""
    from oracle import ref_ops
    B, D, N, Lq = 2, 96, 16, 16384          # ~15 s of host work on the GPU box (7 s at L = 8192)
    g = torch.Generator().manual_seed(0)
    u, z = torch.randn(B, D, Lq, generator=g), torch.randn(B, D, Lq, generator=g)
    delta = 0.5 * torch.rand(B, D, Lq, generator=g)
    A = -0.5 * torch.rand(D, N, generator=g)
    Bm, Cm = torch.randn(B, N, Lq, generator=g), torch.randn(B, N, Lq, generator=g)
    Dv, db = torch.randn(D, generator=g), 0.5 * torch.rand(D, generator=g)
    t0 = time.time()
    with torch.no_grad():
        ref_ops.selective_scan_ref(u, delta, A, Bm, Cm, Dv, z=z, delta_bias=db, delta_softplus=True)
    dt = time.time() - t0
    nbytes = ref_ops.algorithmic_bytes_scan(B, D, Lq, N, 4)
    return {"value": round(nbytes / dt * 1e-9, 4), "unit": "GB/s", "cores": torch.get_num_threads(), "kind": "port",
            "seconds": round(dt, 2),
            "sample": f"oracle selective_scan_ref forward, fp32, B={B} D={D} N={N} L={Lq} (1/16 of the roofline shape's L; "
                      "cost is linear in L), same algorithmic-bytes formula"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("SEGM_FORCE_DDP") == "1"     # the latter: exercise the DDP path on one GPU
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="env://")

    from segmamba_amd.trainer import SyntheticBraTS, build_training_state, train_step
    state = build_training_state(device, distributed, local_rank)
    data = SyntheticBraTS(args.batch, args.size, device, seed=42 + rank)      # trainer.py:331 seeds 42 + rank

    def step():
        image, label = data.next()
        return train_step(state, image, label)

    for _ in range(args.warmup):
        step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        vols = world * args.batch * args.steps
        out = {
            "metric": f"volumes/sec fwd+bwd+step, SegMamba {args.size}^3x4 (whole job; divide by n_gpus for per-GPU)",
            "value": round(vols / elapsed, 4), "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SegMamba(4->4,[2,2,2,2],[48,96,192,384]) train step, {args.batch}x4x{args.size}^3 per GPU, "
                                   "bf16 autocast, CE loss, clip 12, SGD nesterov",
                       "volumes_per_gpu": args.batch, "volume": [args.size] * 3, "parallelism": f"dp{world}",
                       "loss": round(float(loss), 5)},
        }
        if not args.no_roofline:
            out["inference"] = inference_rate(state, device, args.size)
            out["roofline"] = scan_roofline(torch.bfloat16, device)
            out["roofline_fp32"] = scan_roofline(torch.float32, device)
        if not args.no_cpu_baseline and world == 1:       # the host-core baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
