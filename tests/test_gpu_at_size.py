"""-m gpu: the HIP scan kernels against the fp64 C oracle (oracle/scan_ref.c, pinned to the reference's fixtures by
tests/test_oracle_golden.py) AT THE SIZES BASELINE.json names, forward and all eight gradients, at the north-star
tolerances (1e-3 fp32 / 1e-2 bf16; tests/helpers.check_scan):

  stage 0 of config 2 / 3   B=2, D=96,  N=16, L=64^3 = 262144   all three time orders (fp32), slice-interleaved in bf16
  config 1 (one Mamba block, d_model 384)   B=2, D=768, L=262144, bf16      the first, a middle and the last 32-channel tile at the full
                                             length; every channel - dB / dC sum over all 24 d-tiles - at L = 32768 (round 5: the
                                             fp64 oracle over all 768 channels at the full length took 71 s of a suite that has to
                                             stay well inside the driver's 20 minutes; the d-tile sum does not depend on L)
  config 4 (long sequences)  B=1, D=96, bf16, L = 2^21 (everything) and L = 2^24 (16 channels; dB / dC checked at 2^21)

so the carry kernel with 1024+ chunks, the backward beyond L = 4096 and the last steps of the longest sequences are
compared with the oracle, not only checked for finiteness (reference test matrix stops at L = 4096:
mamba/tests/ops/test_selective_scan.py:25).
"""
import time

import pytest
import torch

from oracle import ref_ops, scan_ref
from segmamba_amd import lib as L
from segmamba_amd import ops_raw
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    return L.get_lib()


def _case(B, D, N, Lq, dtype, seed):
    """Reference test distributions (test_selective_scan.py:58-88), generated on the device in the channel-last layout."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(dtype)
    c = {"u": rn(B, Lq, D), "z": rn(B, Lq, D), "g": rn(B, Lq, D),
         "delta": (0.5 * torch.rand(B, Lq, D, device=DEV, generator=g)).to(dtype),
         "A": -0.5 * torch.rand(D, N, device=DEV, generator=g), "B": rn(B, Lq, N), "C": rn(B, Lq, N),
         "D": torch.randn(D, device=DEV, generator=g), "delta_bias": 0.5 * torch.rand(D, device=DEV, generator=g)}
    return c


def _cf(t, order, ns, ch=None):
    """device (B, L, X) -> CPU fp32 (B, X, L) in LOGICAL time order (what the reference's flip / stack copies produce)"""
    if ch is not None:
        t = t[:, :, ch]
    return H.perm(t.transpose(1, 2).contiguous().float().cpu(), order, ns)


def _back(t, order, ns):
    """oracle (B, X, L) logical order -> (B, L, X) physical order"""
    return H.iperm(t, order, ns).transpose(1, 2)


def _run_and_check(hip, c, dtype, order, ns, what, ch=None, want_bc=True, elem_scale=None):
    f = ops_raw.scan_fwd(hip, c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"], True,
                         channel_last=True, time_order=order, nslices=ns, need_out=True, need_ckpt=True, need_last_state=True)
    r = ops_raw.scan_bwd(hip, c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"], c["g"], f["out"],
                         f["ckpt"], True, channel_last=True, time_order=order, nslices=ns, chunk=f["chunk"])
    torch.cuda.synchronize()
    sl = slice(None) if ch is None else ch
    t0 = time.time()
    args = (_cf(c["u"], order, ns, ch), _cf(c["delta"], order, ns, ch), c["A"][sl].cpu(), _cf(c["B"], order, ns),
            _cf(c["C"], order, ns), c["D"][sl].cpu(), _cf(c["z"], order, ns, ch), c["delta_bias"][sl].cpu())
    of = scan_ref.scan_fwd(*args, delta_softplus=True)
    ob = scan_ref.scan_bwd(*args, _cf(c["g"], order, ns, ch), delta_softplus=True, want_bc=want_bc)
    print(f"[{what}] oracle {time.time() - t0:.1f} s on {scan_ref.threads()} threads, chunk {f['chunk']}")
    dev = lambda t: t if ch is None else t[:, :, ch]
    res = {"out": dev(f["out_z"]), "last_state": f["last_state"][:, sl], "du": dev(r["du"]), "ddelta": dev(r["ddelta"]),
           "dz": dev(r["dz"]), "dA": r["dA"][sl], "dD": r["dD"][sl], "ddelta_bias": r["ddelta_bias"][sl]}
    ref = {"out": _back(of["out"], order, ns), "last_state": of["last_state"], "du": _back(ob["du"], order, ns),
           "ddelta": _back(ob["ddelta"], order, ns), "dz": _back(ob["dz"], order, ns), "dA": ob["dA"], "dD": ob["dD"],
           "ddelta_bias": ob["ddelta_bias"]}
    if want_bc:
        res["dB"], res["dC"] = r["dB"], r["dC"]
        ref["dB"], ref["dC"] = _back(ob["dB"], order, ns), _back(ob["dC"], order, ns)
    H.check_scan(res, ref, dtype, what, elem_scale=elem_scale(ref) if elem_scale else 1.0)


@pytest.mark.parametrize("order,dtype", [(L.TIME_FORWARD, torch.float32), (L.TIME_REVERSED, torch.float32),
                                         (L.TIME_INTERLEAVED, torch.float32), (L.TIME_INTERLEAVED, torch.bfloat16),
                                         (L.TIME_INTERLEAVED, torch.float16)])      # fp16: the reference's AMP dtype
def test_stage0_size_forward_and_all_gradients(hip, order, dtype):
    c = _case(2, 96, 16, 64 ** 3, dtype, seed=1 + order)
    _run_and_check(hip, c, dtype, order, 64, f"stage0 L=262144 order={order} {dtype}")


@pytest.mark.parametrize("tile", [0, 11, 23])
def test_config1_block_size(hip, tile):
    c = _case(2, 768, 16, 64 ** 3, torch.bfloat16, seed=7)
    _run_and_check(hip, c, torch.bfloat16, L.TIME_FORWARD, 1, f"config1 D=768 L=262144 bf16 ch {32 * tile}:{32 * tile + 32}",
                   ch=slice(32 * tile, 32 * tile + 32), want_bc=False)


def test_config1_width_every_channel(hip):
    c = _case(2, 768, 16, 32768, torch.bfloat16, seed=8)
    _run_and_check(hip, c, torch.bfloat16, L.TIME_FORWARD, 1, "config1 D=768 L=32768 bf16 (dB / dC over all 24 d-tiles)")


def test_config4_two_million_steps(hip):
    c = _case(1, 96, 16, 1 << 21, torch.bfloat16, seed=21)
    _run_and_check(hip, c, torch.bfloat16, L.TIME_FORWARD, 1, "config4 L=2^21 bf16")


def test_config4_sixteen_million_steps(hip):
    c = _case(1, 96, 16, 1 << 24, torch.bfloat16, seed=24)
    _run_and_check(hip, c, torch.bfloat16, L.TIME_FORWARD, 1, "config4 L=2^24 bf16 ch 32:48", ch=slice(32, 48), want_bc=False)


def test_config4_sixteen_million_steps_fp32_reversed(hip):
    """fp32 I/O at L = 2^24: 6.4 GB per tensor and batch - beyond a 32-bit byte offset from the batch base.  The regular-shape
    kernels address from the lowest row ONE wave touches (scan_fast.h), so only the rows of a wave must fit 32 bits; time-reversed
    so that the descending offsets are exercised at that size too."""
    c = _case(1, 96, 16, 1 << 24, torch.float32, seed=25)
    # |out| reaches 1.3e3 in this case (every other case: order 1 ... 10) and fp32 rounding over 16.7 M steps is relative to the
    # magnitude of the running state, so the absolute term of the 1e-3 bound is scaled by max|out| / 64: 2e-5 of the signal
    # (measured: max |err| 1.2e-2 against the fp64 oracle - bit-identical before and after the round-3 rewrite of the forward row streams,
    # profiles/r03_fp32_2p24_ab.log)
    _run_and_check(hip, c, torch.float32, L.TIME_REVERSED, 1, "config4 L=2^24 fp32 reversed ch 0:16", ch=slice(0, 16), want_bc=False,
                   elem_scale=lambda ref: max(1.0, float(ref["out"].abs().max()) / 64.0))
