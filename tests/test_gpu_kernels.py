"""GPU parity tests: the HIP kernels (through the C ABI of libsegmamba_hip.so) against the CPU oracle and against the
golden fixtures produced by the reference.  Scan tolerances are BASELINE.json's north-star bounds (helpers.check_scan:
|err| <= tol |ref| + tol with tol = 1e-3 fp32 / 1e-2 bf16 - tighter than the reference's own test tolerances); every
comparison logs its margin (helpers.assert_close -> gpurun_out/parity_log.jsonl).  The sizes BASELINE.json names are in
tests/test_gpu_at_size.py (fp64 C oracle)."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from segmamba_amd import lib as L
from segmamba_amd import ops_raw
from oracle import ref_ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    return L.get_lib()          # raises if the HIP library is missing: no fallback


# ---- reference test matrix: mamba/tests/ops/test_selective_scan.py:18-39 ------------------------------------------
@pytest.mark.parametrize("seqlen", [128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("channel_last", [False, True])
def test_scan_reference_matrix(hip, seqlen, groups, channel_last):
    c = H.scan_case(2, 4, 8, seqlen, groups=groups)
    ref = H.scan_oracle(c)
    res = H.run_scan(hip, c, DEV, channel_last)
    H.check_scan(res, ref, torch.float32, f"L={seqlen} G={groups} cl={channel_last}")


@pytest.mark.parametrize("L_", [64, 256])
@pytest.mark.parametrize("G", [1, 2])
def test_scan_golden_from_reference(hip, L_, G):
    """Inputs AND expected outputs/gradients come from the reference's selective_scan_ref (tests/golden)."""
    f = H.load_golden(f"scan_L{L_}_G{G}.npz")
    c = {k: f[k] for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "g")}
    ref = {k: f[k] for k in ("out", "last_state", "du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias")}
    for cl in (False, True):
        H.check_scan(H.run_scan(hip, c, DEV, cl), ref, torch.float32, f"golden L={L_} G={G} cl={cl}")


# ---- SegMamba shapes (dim = 2*d_model, dstate 16), all three time orders, odd lengths, chunk sizes -------------------
@pytest.mark.parametrize("dim,seqlen,ns", [(96, 1536, 64), (192, 512, 32), (384, 256, 16), (768, 64, 8)])
@pytest.mark.parametrize("order", [L.TIME_FORWARD, L.TIME_REVERSED, L.TIME_INTERLEAVED])
def test_scan_segmamba_shapes(hip, dim, seqlen, ns, order):
    c = H.scan_case(2, dim, 16, seqlen, seed=dim)
    ref = H.scan_oracle(c, order, ns)
    res = H.run_scan(hip, c, DEV, True, order, ns)
    H.check_scan(res, ref, torch.float32, f"D={dim} L={seqlen} order={order}")


@pytest.mark.parametrize("seqlen,chunk", [(1, 32), (7, 32), (33, 32), (100, 64), (777, 128), (1000, 256)])
def test_scan_ragged_lengths_and_chunks(hip, seqlen, chunk):
    c = H.scan_case(2, 40, 16, seqlen, seed=seqlen)
    ref = H.scan_oracle(c)
    res = H.run_scan(hip, c, DEV, True, chunk=chunk)
    H.check_scan(res, ref, torch.float32, f"L={seqlen} chunk={chunk}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("channel_last", [False, True])
def test_scan_half_precision(hip, dtype, channel_last):
    c = H.scan_case(2, 96, 16, 512, dtype=dtype)
    ref = H.scan_oracle(c)
    res = H.run_scan(hip, c, DEV, channel_last)
    H.check_scan(res, ref, dtype, f"{dtype}")


@pytest.mark.parametrize("dim,dstate", [(64, 4), (128, 3), (64, 8), (192, 2)])
def test_scan_small_state_wide_channels(hip, dim, dstate):
    c = H.scan_case(2, dim, dstate, 200, seed=dim + dstate)
    H.check_scan(H.run_scan(hip, c, DEV, True), H.scan_oracle(c), torch.float32, f"D={dim} N={dstate}")


def test_scan_optional_arguments(hip):
    for has_z, has_D, has_bias, softplus in [(False, True, True, True), (True, False, False, False), (False, False, False, True)]:
        c = H.scan_case(1, 32, 16, 200, has_z=has_z, has_D=has_D, has_bias=has_bias, seed=3)
        ref = H.scan_oracle(c, softplus=softplus)
        res = H.run_scan(hip, c, DEV, True, softplus=softplus)
        H.check_scan(res, ref, torch.float32, f"z={has_z} D={has_D} bias={has_bias} softplus={softplus}")


def test_scan_forward_is_deterministic_and_chunk_invariant(hip):
    """Bit-identical repeats (reference: test_causal_conv1d_race_condition idea, applied to the scan forward);
    different chunkings agree to rounding."""
    c = H.scan_case(2, 96, 16, 4096, seed=11)
    a = H.run_scan(hip, c, DEV, True, chunk=128, backward=False)["out"]
    for _ in range(5):
        b = H.run_scan(hip, c, DEV, True, chunk=128, backward=False)["out"]
        assert torch.equal(a, b)
    d = H.run_scan(hip, c, DEV, True, chunk=1024, backward=False)["out"]
    H.assert_close(d, a, 1e-4, 1e-5 * float(a.abs().max()), "chunk 1024 vs 128")      # fp32 rounding, different grouping


def test_scan_full_size_properties(hip):
    """BASELINE config-1 size per direction (B=2, D=768, L=64^3 is 3.2 GB fp32 per tensor: run the stage-0 SegMamba size
    B=2, D=96, L=262144 instead) through size-independent properties: linearity in u, reversal symmetry, and a
    64-row spot check against the oracle."""
    B_, D_, N_, L_ = 2, 96, 16, 262144
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    u1, u2 = r(B_, L_, D_), r(B_, L_, D_)
    delta = 0.5 * torch.rand(B_, L_, D_, device=DEV, generator=g)
    A = -0.5 * torch.rand(D_, N_, device=DEV, generator=g)
    Bm, Cm = r(B_, L_, N_), r(B_, L_, N_)
    Dv, db = r(D_), 0.5 * torch.rand(D_, device=DEV, generator=g)

    def fwd(u, order=L.TIME_FORWARD, dl=delta, b=Bm, c=Cm):
        return ops_raw.scan_fwd(hip, u, dl, A, b, c, Dv, None, db, True, channel_last=True, time_order=order)["out"]

    y1, y2, y12 = fwd(u1), fwd(u2), fwd(u1 + 2.0 * u2)
    H.assert_close(y12, y1 + 2.0 * y2, 2e-3, 2e-3, "linearity in u")
    # REVERSED order on explicitly flipped inputs == FORWARD order, flipped
    yr = fwd(u1.flip(1), L.TIME_REVERSED, delta.flip(1), Bm.flip(1), Cm.flip(1))
    H.assert_close(yr.flip(1), y1, 1e-4, 1e-4, "reversal symmetry")
    # spot rows: the first 2048 steps depend only on the first 2048 inputs
    T = 2048
    ref = ref_ops.selective_scan_ref(u1[:, :T].transpose(1, 2).cpu(), delta[:, :T].transpose(1, 2).cpu(), A.cpu(),
                                     Bm[:, :T].transpose(1, 2).cpu(), Cm[:, :T].transpose(1, 2).cpu(), Dv.cpu(),
                                     delta_bias=db.cpu(), delta_softplus=True)
    H.assert_close(y1[:, :T].transpose(1, 2), ref, 6e-4, 2e-3, "prefix vs oracle")


@pytest.mark.parametrize("log2_len", [21, 24])
def test_scan_long_sequence_stress(hip, log2_len):
    """BASELINE config 4: B=1, D=96, N=16, bf16, L = 2^21 (what the reference stem yields for a 256^3 volume) and
    L = 2^24 = 16.7 M (the figure BASELINE.json quotes; the largest L the 24-bit time indices take).  No CPU oracle can
    walk these lengths, so: the first 2048 steps against the oracle (the scan is causal), determinism, chunk-length
    invariance of the forward, and a finite, repeatable backward."""
    B_, D_, N_, L_ = 1, 96, 16, 1 << log2_len
    g = torch.Generator(device=DEV).manual_seed(log2_len)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).bfloat16()
    u, z, dout = r(B_, L_, D_), r(B_, L_, D_), r(B_, L_, D_)
    delta = (0.5 * torch.rand(B_, L_, D_, device=DEV, generator=g)).bfloat16()
    A = -0.5 * torch.rand(D_, N_, device=DEV, generator=g)
    Bm, Cm = r(B_, L_, N_), r(B_, L_, N_)
    Dv, db = torch.randn(D_, device=DEV, generator=g), 0.5 * torch.rand(D_, device=DEV, generator=g)

    def fwd(chunk=0):
        return ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, chunk=chunk, need_out=True,
                                need_ckpt=True, need_last_state=True)
    f = fwd()
    assert torch.isfinite(f["out_z"].float()).all() and torch.isfinite(f["last_state"]).all()
    f2 = fwd()
    assert torch.equal(f["out_z"], f2["out_z"]) and torch.equal(f["last_state"], f2["last_state"])
    fc = fwd(chunk=2 * f["chunk"])
    assert (fc["out_z"].float() - f["out_z"].float()).abs().max() <= 2e-2 * float(f["out_z"].float().abs().max())
    T = 2048
    cpu = lambda t: t[:, :T].transpose(1, 2).float().cpu()
    ref = ref_ops.selective_scan_ref(cpu(u), cpu(delta), A.cpu(), cpu(Bm), cpu(Cm), Dv.cpu(), z=cpu(z), delta_bias=db.cpu(),
                                     delta_softplus=True)
    H.assert_close(f["out_z"][:, :T].transpose(1, 2).float(), ref, 3e-2, 5e-2, "prefix vs oracle")
    bw = lambda: ops_raw.scan_bwd(hip, u, delta, A, Bm, Cm, Dv, z, db, dout, f["out"], f["ckpt"], True, channel_last=True,
                                  chunk=f["chunk"])
    g1, g2 = bw(), bw()
    for k in ("du", "ddelta", "dz", "dA", "dD", "ddelta_bias"):
        assert torch.isfinite(g1[k].float()).all(), k
        assert torch.equal(g1[k], g2[k]), k
    assert torch.isfinite(g1["dB"]).all() and torch.isfinite(g1["dC"]).all()


# ---- causal conv1d: reference matrix causal-conv1d/tests/test_causal_conv1d.py:14-75 (dim reduced) -------------------
@pytest.mark.parametrize("seqlen", [8, 16, 32, 64, 128, 151, 256, 372, 512, 784, 1024, 1134, 2048, 4096])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16])
def test_causal_conv1d_reference_matrix(hip, seqlen, width, itype):
    rtol, atol = (3e-4, 1e-3) if itype == torch.float32 else (3e-3, 5e-3)
    if itype == torch.bfloat16:
        rtol, atol = 1e-2, 5e-2
    torch.manual_seed(0)
    dim = 96 + 32
    for channel_last in (False, True):
        for silu, has_bias in ((True, True), (False, False)):
            # non-contiguous batch / channel strides: a slice of a larger tensor (reference :42)
            if channel_last:
                big = torch.randn(2, seqlen, 64 + dim + 32, device=DEV, dtype=itype)
                x = big[:, :, 64:64 + dim]
            else:
                big = torch.randn(2, 64 + dim + 32, seqlen, device=DEV, dtype=itype)
                x = big[:, 64:64 + dim, :]
            w = torch.randn(dim, width, device=DEV)
            b = torch.randn(dim, device=DEV) if has_bias else None
            xr = (x.transpose(1, 2) if channel_last else x).detach().float().cpu().requires_grad_()
            wr = w.cpu().requires_grad_()
            br = b.cpu().requires_grad_() if has_bias else None
            ref = ref_ops.causal_conv1d_ref(xr, wr, br, "silu" if silu else None)
            g = torch.randn(2, dim, seqlen).to(itype).float()       # same rounded upstream gradient on both sides
            ref.backward(g)
            out = ops_raw.conv1d_fwd(hip, x, w, b, silu, channel_last=channel_last)
            gd = (g.transpose(1, 2) if channel_last else g).to(DEV, itype).contiguous()
            dx, dw, dbias = ops_raw.conv1d_bwd(hip, x, w, b, gd, silu, channel_last=channel_last)
            tr = (lambda t: t.transpose(1, 2)) if channel_last else (lambda t: t)
            what = f"conv L={seqlen} W={width} {itype} cl={channel_last} silu={silu}"
            H.assert_close(tr(out), ref.to(itype), rtol, atol, what + " out")
            H.assert_close(tr(dx), xr.grad.to(itype), rtol * 3, atol * 3, what + " dx")
            # weight / bias gradients are fp32 sums of products of the SAME rounded operands on both sides: only the summation
            # order differs.  16-bit: rtol 1e-2 + 5e-3 max|ref| (<= 1e-2 max|ref|; the reference's 1e-3 compares two
            # 16-bit pipelines with each other, causal-conv1d/tests/test_causal_conv1d.py:34,73-75)
            wtol = (1e-3, 1e-3 * max(1.0, float(wr.grad.abs().max()))) if itype == torch.float32 else (1e-2, 5e-3 * float(wr.grad.abs().max()))
            H.assert_close(dw, wr.grad, wtol[0], wtol[1], what + " dweight")
            if has_bias:
                btol = (1e-3, 1e-3 * max(1.0, float(br.grad.abs().max()))) if itype == torch.float32 else (1e-2, 5e-3 * float(br.grad.abs().max()))
                H.assert_close(dbias, br.grad, btol[0], btol[1], what + " dbias")


@pytest.mark.parametrize("order,ns", [(L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 8), (L.TIME_INTERLEAVED, 64)])
def test_causal_conv1d_time_orders(hip, order, ns):
    torch.manual_seed(1)
    x = torch.randn(2, 96, 1024)
    w, b, g = torch.randn(96, 4), torch.randn(96), torch.randn(2, 96, 1024)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = H.iperm(ref_ops.causal_conv1d_ref(H.perm(xr, order, ns), wr, br, "silu"), order, ns)
    ref.backward(g)
    xd, gd = x.transpose(1, 2).contiguous().to(DEV), g.transpose(1, 2).contiguous().to(DEV)
    out = ops_raw.conv1d_fwd(hip, xd, w.to(DEV), b.to(DEV), True, channel_last=True, time_order=order, nslices=ns)
    dx, dw, db = ops_raw.conv1d_bwd(hip, xd, w.to(DEV), b.to(DEV), gd, True, channel_last=True, time_order=order, nslices=ns)
    H.assert_close(out.transpose(1, 2), ref, 3e-4, 1e-3, "out")
    H.assert_close(dx.transpose(1, 2), xr.grad, 3e-4, 1e-3, "dx")
    H.assert_close(dw, wr.grad, 1e-3, 1e-2, "dw")
    H.assert_close(db, br.grad, 1e-3, 1e-2, "db")


def test_causal_conv1d_repeatability(hip):
    """reference test_causal_conv1d_race_condition (:117-173): out and dx bit-identical across repeats; here dweight /
    dbias are too, because the per-chunk partials are reduced in a fixed order (no atomics)."""
    torch.manual_seed(0)
    x = torch.randn(2, 2048, 4096 + 32, device=DEV, dtype=torch.bfloat16)
    w, b = torch.randn(4096 + 32, 4, device=DEV), torch.randn(4096 + 32, device=DEV)
    g = torch.randn_like(x)
    out0 = ops_raw.conv1d_fwd(hip, x, w, b, True, channel_last=True)
    dx0, dw0, db0 = ops_raw.conv1d_bwd(hip, x, w, b, g, True, channel_last=True)
    for _ in range(50):
        out = ops_raw.conv1d_fwd(hip, x, w, b, True, channel_last=True)
        dx, dw, db = ops_raw.conv1d_bwd(hip, x, w, b, g, True, channel_last=True)
        assert torch.equal(out, out0) and torch.equal(dx, dx0) and torch.equal(dw, dw0) and torch.equal(db, db0)


@pytest.mark.parametrize("order,ns", [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)])
def test_scan_with_dt_proj_inside_the_launch(hip, order, ns):
    """Round 4 (north star N1, opt-in): delta = W_dt . x_dbl[:, :R] formed inside the forward scan passes at SegMamba's stage-0
    row layout (R = 3 of 40-column x_dbl rows, 96 channels).  The delta the apply pass writes is the fp64 product rounded to bf16
    within one ulp; the scan over that written delta equals the fused launch bit for bit (out, out_z, checkpoints)."""
    torch.manual_seed(3)
    B, Lq, D, N, R, P8 = 2, 64 * 256, 96, 16, 3, 40
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, device=DEV).to(dt)
    u, z = rn(B, Lq, D), rn(B, Lq, D)
    x_dbl = rn(B, Lq, P8)
    w = (0.3 * torch.randn(D, R, device=DEV)).to(dt).float().contiguous()
    A = -0.5 * torch.rand(D, N, device=DEV)
    Dv, db = torch.randn(D, device=DEV), 0.5 * torch.rand(D, device=DEV)
    Bm, Cm = x_dbl[:, :, 4:4 + N], x_dbl[:, :, 4 + N:4 + 2 * N]
    kw = dict(channel_last=True, time_order=order, nslices=ns, chunk=256, need_out=True, need_ckpt=True)
    assert ops_raw.scan_fused_conv_supported(hip, B, D, Lq, ns, order, 256)          # the regular-shape kernels take it
    delta = torch.full((B, Lq, D), float("nan"), device=DEV, dtype=dt)
    fused = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, dt_x=x_dbl[:, :, :R], dt_weight=w, **kw)
    exact = x_dbl[:, :, :R].double() @ w.double().t()
    assert torch.isfinite(delta.float()).all()
    assert ((delta.double() - exact).abs() <= 2.0 ** -8 * exact.abs() + 1e-6).all()
    ref = ops_raw.scan_fwd(hip, u, delta.clone(), A, Bm, Cm, Dv, z, db, True, **kw)
    for k in ("out", "out_z", "ckpt"):
        assert torch.equal(fused[k], ref[k]), k


def test_selective_scan_backward_repeatability(hip):
    """Round 4: dB / dC are sums over the d-tiles in a FIXED order (per-tile fp32 slabs + a sum kernel, csrc/scan_bwd_w8.hip) - no
    float atomics: every gradient of the SegMamba stage-0 shape (three d-tiles of 32 channels) is bit-identical across 20
    launches; and dB / dC written in the tensors' own type into strided column windows equal the fp32 results rounded once."""
    torch.manual_seed(1)
    B, Lq, D, N = 2, 4096, 96, 16
    rn = lambda *s: torch.randn(*s, device=DEV)
    dt = torch.bfloat16
    u, z, g = rn(B, Lq, D).to(dt), rn(B, Lq, D).to(dt), rn(B, Lq, D).to(dt)
    delta = (0.5 * torch.rand(B, Lq, D, device=DEV)).to(dt)
    A = -0.5 * torch.rand(D, N, device=DEV) - 0.05
    Bm, Cm = rn(B, Lq, N).to(dt), rn(B, Lq, N).to(dt)
    Dv, db = rn(D), 0.5 * torch.rand(D, device=DEV)
    for order, ns in ((L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)):
        # chunk 256 as at the stage-0 length (the default for L = 4096 is 16 steps: shorter than the 64 slices, and an interleaved
        # order is only a regular shape when whole slice rounds fit a chunk - the general kernels accumulate dB / dC atomically)
        f = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=True, time_order=order, nslices=ns,
                             chunk=256, need_out=True, need_ckpt=True)
        assert f["chunk"] == 256 and hip.dll.segm_selective_scan_regular_shape(B, D, N, Lq, 256, order, ns) == 1
        kw = dict(channel_last=True, time_order=order, nslices=ns, chunk=f["chunk"])
        args = (u, delta, A, Bm, Cm, Dv, z, db, g, f["out"], f["ckpt"], True)
        r0 = ops_raw.scan_bwd(hip, *args, **kw)
        assert not r0["dbc_native"]
        for _ in range(20):
            r = ops_raw.scan_bwd(hip, *args, **kw)
            for k in ("du", "ddelta", "dz", "dA", "dB", "dC", "dD", "ddelta_bias"):
                assert torch.equal(r[k], r0[k]), (k, order)
        wide = torch.zeros(B, Lq, 40, device=DEV, dtype=dt)
        r = ops_raw.scan_bwd(hip, *args, dB=wide[:, :, 4:20], dC=wide[:, :, 20:36], **kw)
        assert r["dbc_native"]
        assert torch.equal(wide[:, :, 4:20], r0["dB"].to(dt)) and torch.equal(wide[:, :, 20:36], r0["dC"].to(dt))
        assert not bool(wide[:, :, :4].any()) and not bool(wide[:, :, 36:].any())


def test_selective_scan_backward_native_dbc_channel_first(hip):
    """the layouts the 8-step-window backward newly accepts - channel-FIRST rows (stride_d != 1: the reference's own
    (B, D, L) layout) with time-fastest B / C (B, N, L) - through the native dB / dC path (dB / dC written once, in the tensors' own
    16-bit type): equal to the fp32 results rounded once, and every gradient equal to the channel-last launch's on the same data."""
    torch.manual_seed(3)
    B, Lq, D, N = 2, 2048, 64, 16
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, device=DEV)
    u, z, g = rn(B, D, Lq).to(dt), rn(B, D, Lq).to(dt), rn(B, D, Lq).to(dt)
    delta = (0.5 * torch.rand(B, D, Lq, device=DEV)).to(dt)
    A = -0.5 * torch.rand(D, N, device=DEV) - 0.05
    Bm, Cm = rn(B, N, Lq).to(dt), rn(B, N, Lq).to(dt)
    Dv, db = rn(D), 0.5 * torch.rand(D, device=DEV)
    f = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, channel_last=False, chunk=256, need_out=True, need_ckpt=True)
    args = (u, delta, A, Bm, Cm, Dv, z, db, g, f["out"], f["ckpt"], True)
    r0 = ops_raw.scan_bwd(hip, *args, channel_last=False, chunk=f["chunk"])
    dBn, dCn = torch.zeros(B, N, Lq, device=DEV, dtype=dt), torch.zeros(B, N, Lq, device=DEV, dtype=dt)
    r = ops_raw.scan_bwd(hip, *args, channel_last=False, chunk=f["chunk"], dB=dBn, dC=dCn)
    if hip.dll.segm_selective_scan_regular_shape(B, D, N, Lq, f["chunk"], L.TIME_FORWARD, 1) == 1:
        assert r["dbc_native"] and not r0["dbc_native"]
        assert torch.equal(dBn, r0["dB"].reshape(B, N, Lq).to(dt)) and torch.equal(dCn, r0["dC"].reshape(B, N, Lq).to(dt))
    else:
        assert not r["dbc_native"]
    for k in ("du", "ddelta", "dz", "dA", "dD", "ddelta_bias"):
        assert torch.equal(r[k], r0[k]), k
    # the same data channel-last: same arithmetic per (b, d) row and state, same d-tile summation order
    tl = lambda t: t.transpose(1, 2).contiguous()
    fl = ops_raw.scan_fwd(hip, tl(u), tl(delta), A, tl(Bm), tl(Cm), Dv, tl(z), db, True, channel_last=True, chunk=256, need_out=True, need_ckpt=True)
    rl = ops_raw.scan_bwd(hip, tl(u), tl(delta), A, tl(Bm), tl(Cm), Dv, tl(z), db, tl(g), fl["out"], fl["ckpt"], True, channel_last=True, chunk=fl["chunk"])
    for k in ("du", "ddelta", "dz"):
        assert (tl(r0[k]).float() - rl[k].float()).abs().max() <= 1e-2 * rl[k].float().abs().max(), k
    for k in ("dA", "dD", "ddelta_bias"):
        assert (r0[k] - rl[k]).abs().max() <= 1e-3 * rl[k].abs().max() + 1e-4, k


@pytest.mark.parametrize("variant,S,bias", [("chain48", 128, 0.0), ("chain48", 128, 3.0), ("chain32", 64, 0.0), ("chain32", 32, 2.0)])
def test_conv_statistics_epilogue_against_fp64(hip, variant, S, bias):
    """round 5: InstanceNorm statistics summed in the convolution's epilogue, at 2 M voxels per instance
    (128^3; and the 32-wide kernel's shapes), against fp64 statistics of the volume the convolution stored: the mean within 2e-4 of
    the standard deviation, 1 / std within 2e-4 relative - also with a mean three standard deviations off zero (a bias), where a
    sum-of-squares form loses the most; the normalised output equals the library's own-pass result within one bf16 step."""
    torch.manual_seed(S + int(bias))
    x = torch.randn(2, 48, S, S, S, device=DEV).bfloat16()
    w = (0.04 * torch.randn(48, 48, 3, 3, 3, device=DEV)).bfloat16()
    b = torch.full((48,), bias, device=DEV) if bias else None
    wp = ops_raw.pack_conv3d_weight(w)
    kw = dict(chain=True, pitch48=True) if variant == "chain48" else dict(chain32=True)
    y, stats = ops_raw.conv3d_k3_fwd(hip, x, wp, b, want_stats=True, **kw)
    assert stats is not None and torch.equal(y, ops_raw.conv3d_k3_fwd(hip, x, wp, b, **kw))
    assert float(stats[..., 0].sum(-1).min()) == float(stats[..., 0].sum(-1).max()) == float(S ** 3)
    yd = y.double()
    mean64, var64 = yd.mean((2, 3, 4)).flatten(), yd.var((2, 3, 4), unbiased=False).flatten()
    y1, m1, r1 = ops_raw.instnorm_fwd(hip, y, None, "leaky_relu", 0.01, 1e-5, stats=stats)
    y0, m0, r0 = ops_raw.instnorm_fwd(hip, y, None, "leaky_relu", 0.01, 1e-5)
    std64 = (var64 + 1e-5).sqrt()
    assert ((m1.double() - mean64).abs() / std64).max() <= 2e-4, float(((m1.double() - mean64).abs() / std64).max())
    assert (r1.double() * std64 - 1).abs().max() <= 2e-4, float((r1.double() * std64 - 1).abs().max())
    assert (y1.float() - y0.float()).abs().max() <= 2.0 ** -7 * max(1.0, float(y0.float().abs().max()))


def test_error_behaviour(hip):
    """reference TORCH_CHECKs (selective_scan.cpp:233-303, causal_conv1d.cpp:136-170) surface as RuntimeError."""
    x = torch.randn(1, 8, 16, device=DEV)
    with pytest.raises(RuntimeError):
        ops_raw.conv1d_fwd(hip, x, torch.randn(8, 5, device=DEV), None, True)            # width 5
    with pytest.raises(RuntimeError):
        ops_raw.conv1d_fwd(hip, x.double(), torch.randn(8, 4, device=DEV), None, True)   # dtype
    c = H.scan_case(1, 8, 17, 16)
    with pytest.raises(RuntimeError):
        H.run_scan(hip, c, DEV, False, backward=False)                                    # dstate 17
    c = H.scan_case(1, 8, 4, 16)
    c["delta"] = c["delta"][:, :, :8]
    with pytest.raises(RuntimeError):
        H.run_scan(hip, c, DEV, False, backward=False)                                    # shape mismatch


# ---- 3x3x3 weight gradient (stem / decoder convolutions): against aten.convolution_backward -----------------------
def _aten_wgrad(x, dy):
    w = torch.empty(dy.shape[1], x.shape[1], 3, 3, 3, device=x.device, dtype=x.dtype)
    return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1,
                                               [False, True, False])[1]


@pytest.mark.parametrize("shape", [(1, 48, 48, 8, 8, 32), (2, 48, 96, 5, 7, 64), (2, 96, 48, 16, 16, 32),
                                   (1, 48, 48, 32, 32, 128), (2, 96, 48, 8, 8, 8), (1, 48, 96, 6, 16, 16), (1, 48, 48, 3, 9, 40)])
def test_conv3d_k3_wgrad_matches_fp32_convolution_backward(hip, shape):
    B, cin, cout, D, H_, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(B, cin, D, H_, W, device=DEV, generator=g).bfloat16()
    dy = torch.randn(B, cout, D, H_, W, device=DEV, generator=g).bfloat16()
    ref = _aten_wgrad(x.float(), dy.float())                  # fp32 reference on the same bf16-rounded inputs
    dw = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
    # products of bf16 values are exact in fp32; only the fp32 summation order differs
    assert (dw - ref).abs().max() <= 2e-5 * ref.abs().max() * max(1.0, (B * D * H_ * W / 4096) ** 0.5)
    assert torch.equal(ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32), dw)       # deterministic
    assert torch.equal(ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.bfloat16), dw.bfloat16())


def test_conv3d_k3_wgrad_channel_slices(hip):
    g = torch.Generator(device=DEV).manual_seed(11)
    xb = torch.randn(2, 96, 8, 8, 64, device=DEV, generator=g).bfloat16()
    dyb = torch.randn(2, 96, 8, 8, 64, device=DEV, generator=g).bfloat16()
    xs, dys = xb[:, 48:], dyb[:, :48]
    ref = _aten_wgrad(xs.float().contiguous(), dys.float().contiguous())
    dw = ops_raw.conv3d_k3_wgrad(hip, xs, dys, torch.float32)
    assert (dw - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_conv3d_same_autograd_with_mfma_wgrad(hip, monkeypatch):
    """the dispatcher's backward with the library kernel forced in equals torch's own conv3d backward."""
    from segmamba_amd import conv3d as C3
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(1, 48, 8, 16, 32, device=DEV, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=DEV, generator=g)).bfloat16().requires_grad_()
    dy = torch.randn(1, 48, 8, 16, 32, device=DEV, generator=g).bfloat16()
    # the last candidates: the chained forward kernel (forward / data gradient) and the MFMA weight-gradient kernel
    # (the two untimed forward variants are candidates only with SEGM_CONV_FWD_UNTIMED=1: tests/test_zz_gpu_unmeasured.py)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    y = C3.conv3d_same(x, w)
    gx, gw = torch.autograd.grad(y, (x, w), dy)
    x2, w2 = x.detach().float().requires_grad_(), w.detach().float().requires_grad_()
    y2 = torch.nn.functional.conv3d(x2, w2, None, 1, 1)
    gx2, gw2 = torch.autograd.grad(y2, (x2, w2), dy.float())
    assert (y.float() - y2).abs().max() <= 2e-2 * y2.abs().max()
    assert (gw.float() - gw2).abs().max() <= 1e-2 * gw2.abs().max()
    assert (gx.float() - gx2).abs().max() <= 2e-2 * gx2.abs().max()


# ---- InstanceNorm3d (+ residual) (+ activation): against the ATen chain the reference runs --------------------------
def _aten_norm_chain(x, res, act, slope):
    import torch.nn.functional as F
    y = F.instance_norm(x, eps=1e-5)
    if res is not None:
        y = y + res
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky_relu":
        y = F.leaky_relu(y, slope)
    return y


@pytest.mark.parametrize("shape", [(2, 48, 32, 32, 32), (1, 96, 16, 16, 16), (2, 384, 8, 8, 8), (1, 3, 5, 7, 9)])
@pytest.mark.parametrize("act,with_res", [("none", False), ("relu", False), ("leaky_relu", False), ("leaky_relu", True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_instance_norm_act_matches_aten(hip, shape, act, with_res, dtype):
    from segmamba_amd import fused_norm
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = (1.5 * torch.randn(shape, device=DEV, generator=g) + 0.3).to(dtype)
    res = torch.randn(shape, device=DEV, generator=g).to(dtype) if with_res else None
    gy = torch.randn(shape, device=DEV, generator=g).to(dtype)
    xr = x.float().requires_grad_()
    rr = res.float().requires_grad_() if with_res else None
    ref = _aten_norm_chain(xr, rr, act, 0.01)                      # fp32 reference on the same (rounded) inputs
    gref = torch.autograd.grad(ref, (xr, rr) if with_res else (xr,), gy.float())
    xk = x.clone().requires_grad_()
    rk = res.clone().requires_grad_() if with_res else None
    y = fused_norm.instance_norm_act(xk, act, 0.01, 1e-5, rk)
    gk = torch.autograd.grad(y, (xk, rk) if with_res else (xk,), gy)
    tol = 1e-4 if dtype == torch.float32 else 2e-2              # north star: 1e-3 fp32 / 1e-2 bf16 (relative to O(1..5) values)
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    # away from the activation kink (|pre-activation| tiny) the gradients agree; at the kink rounding may flip the mask
    pre = torch.nn.functional.instance_norm(x.float(), eps=1e-5) + (res.float() if with_res else 0)
    safe = (pre.abs() > 2e-2) if dtype != torch.float32 else (pre.abs() > 1e-5)
    if act == "none" or dtype == torch.float32:
        assert (gk[0].float() - gref[0]).abs().max() <= 10 * tol * max(1.0, float(gref[0].abs().max()))
    if with_res:
        assert ((gk[1].float() - gref[1]).abs() * safe).max() <= tol * max(1.0, float(gref[1].abs().max()))


def test_instance_norm_full_size_properties(hip):
    """BASELINE size (2 x 48 x 128^3, bf16): zero mean / unit variance per instance, determinism, dx orthogonality."""
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (2.0 * torch.randn(2, 48, 128, 128, 128, device=DEV, generator=g) - 0.7).bfloat16()
    y, mean, rstd = ops_raw.instnorm_fwd(hip, x, None, "none")
    yf = y.float().flatten(2)
    assert yf.mean(-1).abs().max() < 2e-3 and (yf.var(-1, unbiased=False) - 1).abs().max() < 5e-3
    y2, mean2, rstd2 = ops_raw.instnorm_fwd(hip, x, None, "none")
    assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    dy = torch.randn(x.shape, device=DEV, generator=g).bfloat16()
    dx, _ = ops_raw.instnorm_bwd(hip, x, dy, mean, rstd)
    dxf = dx.float().flatten(2)
    assert dxf.mean(-1).abs().max() < 2e-3                         # dx is orthogonal to 1 ...
    assert (dxf * yf).mean(-1).abs().max() < 5e-3                  # ... and to xhat


# ---- channel-first <-> channel-last transposes ----------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 48, 32768), (2, 4096, 384), (1, 130, 72), (3, 7, 13), (2, 768, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_add_bit_exact(hip, shape, dtype):
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(shape, device=DEV, generator=g).to(dtype)
    add = torch.randn(shape[0], shape[2], shape[1], device=DEV, generator=g).to(dtype)
    assert torch.equal(ops_raw.transpose_add(hip, x), x.transpose(1, 2).contiguous())           # a pure move: bit exact
    assert torch.equal(ops_raw.transpose_add(hip, x, add), (x.transpose(1, 2).float() + add.float()).to(dtype))


def test_token_layout_autograd_full_size(hip):
    from segmamba_amd import layout
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(2, 48, 128, 128, 128, device=DEV, generator=g).bfloat16().requires_grad_()
    tok = layout.volume_to_tokens(x)
    assert tok.shape == (2, 128 ** 3, 48) and torch.equal(tok, x.detach().flatten(2).transpose(1, 2))
    w = torch.randn(2, 128 ** 3, 48, device=DEV, generator=g).bfloat16()
    y = layout.tokens_to_volume_add(tok * w, x)                     # round trip + skip
    gy = torch.randn(y.shape, device=DEV, generator=g).bfloat16()
    (gx,) = torch.autograd.grad(y, x, gy)
    wv = w.transpose(1, 2).reshape(x.shape)
    assert torch.equal(y, (x.detach() * wv).float().add(x.detach().float()).bfloat16())
    want = gy.float() * wv.float() + gy.float()
    assert ((gx.float() - want).abs() <= 2.0 ** -6 * (gy.float() * wv.float()).abs() + 2.0 ** -7 * want.abs() + 1e-6).all()


# ---- 3x3x3 forward / data-gradient kernel: against fp32 conv3d on the same bf16-rounded operands -----------------------
@pytest.mark.parametrize("shape", [(1, 48, 8, 8, 32), (2, 32, 5, 7, 64), (2, 48, 16, 16, 128), (1, 96, 6, 16, 16),
                                   (1, 16, 3, 9, 72), (2, 48, 2, 20, 8)])
def test_conv3d_k3_fwd_matches_fp32_conv(hip, shape):
    B, cout, D, H_, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(B, 48, D, H_, W, device=DEV, generator=g).bfloat16()
    w = (0.05 * torch.randn(cout, 48, 3, 3, 3, device=DEV, generator=g)).bfloat16()
    bias = torch.randn(cout, device=DEV, generator=g)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    y = ops_raw.conv3d_k3_fwd(hip, x, ops_raw.pack_conv3d_weight(w), bias)
    assert (y.float() - ref).abs().max() <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))       # one bf16 rounding of the result
    assert torch.equal(ops_raw.conv3d_k3_fwd(hip, x, ops_raw.pack_conv3d_weight(w), bias), y)   # deterministic
    xs = torch.randn(B, 96, D, H_, W, device=DEV, generator=g).bfloat16()[:, 48:]               # channel slice (strided view)
    ref = torch.nn.functional.conv3d(xs.float(), w.float(), None, 1, 1)
    y = ops_raw.conv3d_k3_fwd(hip, xs, ops_raw.pack_conv3d_weight(w))
    assert (y.float() - ref).abs().max() <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(1, 48, 8, 8, 32), (2, 48, 16, 16, 128), (1, 96, 6, 16, 16), (2, 48, 2, 20, 8), (1, 48, 3, 40, 72)])
def test_conv3d_k3_fwd_chained_k_parts(hip, shape):
    """SEGM_CONV_FWD_CHAIN (K parts pipelined through LDS, 160 KB of LDS per workgroup) and SEGM_CONV_FWD_ACCUMULATE
    (a second 48-channel input block added in place) against fp32 conv3d and against the default kernel."""
    B, cout, D, H_, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape) + 5)
    x = torch.randn(B, 96, D, H_, W, device=DEV, generator=g).bfloat16()
    w = (0.05 * torch.randn(cout, 96, 3, 3, 3, device=DEV, generator=g)).bfloat16()
    bias = torch.randn(cout, device=DEV, generator=g)
    w0, w1 = ops_raw.pack_conv3d_weight(w[:, :48]), ops_raw.pack_conv3d_weight(w[:, 48:])
    ref0 = torch.nn.functional.conv3d(x[:, :48].float(), w[:, :48].float(), bias, 1, 1)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    tol = 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    y = ops_raw.conv3d_k3_fwd(hip, x[:, :48], w0, bias, chain=True)
    assert (y.float() - ref0).abs().max() <= tol
    assert torch.equal(ops_raw.conv3d_k3_fwd(hip, x[:, :48], w0, bias, chain=True), y)          # deterministic
    yd = ops_raw.conv3d_k3_fwd(hip, x[:, :48], w0, bias)
    assert (y.float() - yd.float()).abs().max() <= tol                                           # same sums, other order
    for kw in (dict(chain=True), dict()):
        acc = yd.clone()
        ops_raw.conv3d_k3_fwd(hip, x[:, 48:], w1, None, out=acc, accumulate=True, **kw)
        assert (acc.float() - ref).abs().max() <= 2 * tol


def test_conv3d_same_autograd_with_library_kernels(hip, monkeypatch):
    """the dispatcher with the library's fwd / dgrad / wgrad kernels forced in == torch's conv3d autograd (fp32)."""
    from segmamba_amd import conv3d as C3
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(2, 48, 8, 16, 32, device=DEV, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=DEV, generator=g)).bfloat16().requires_grad_()
    bias = torch.randn(48, device=DEV, generator=g).bfloat16().requires_grad_()
    dy = torch.randn(2, 48, 8, 16, 32, device=DEV, generator=g).bfloat16()
    # the last candidates: the chained forward kernel (forward / data gradient) and the MFMA weight-gradient kernel
    # (the two untimed forward variants are candidates only with SEGM_CONV_FWD_UNTIMED=1: tests/test_zz_gpu_unmeasured.py)
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[-1]())
    y = C3.conv3d_same(x, w, bias)
    gx, gw, gb = torch.autograd.grad(y, (x, w, bias), dy)
    x2, w2, b2 = (t.detach().float().requires_grad_() for t in (x, w, bias))
    y2 = torch.nn.functional.conv3d(x2, w2, b2, 1, 1)
    gx2, gw2, gb2 = torch.autograd.grad(y2, (x2, w2, b2), dy.float())
    for got, want in ((y, y2), (gx, gx2), (gw, gw2), (gb, gb2)):
        assert (got.float() - want).abs().max() <= 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dim,seqlen,order,ns,dtype", [(96, 4096, L.TIME_FORWARD, 1, torch.bfloat16),
                                                       (96, 4096, L.TIME_INTERLEAVED, 64, torch.float32),
                                                       (192, 2048, L.TIME_REVERSED, 1, torch.float32),
                                                       (384, 512, L.TIME_INTERLEAVED, 16, torch.bfloat16)])
def test_scan_regular_and_general_kernels_agree(hip, monkeypatch, dim, seqlen, order, ns, dtype):
    """the regular-shape kernels (scan_*_fast.hip) and the general ones compute the same forward and gradients."""
    c = H.scan_case(2, dim, 16, seqlen, dtype=dtype, seed=dim + seqlen)
    monkeypatch.delenv("SEGM_SCAN_FAST", raising=False)
    fast = H.run_scan(hip, c, DEV, True, order, ns)
    monkeypatch.setenv("SEGM_SCAN_FAST", "0")
    slow = H.run_scan(hip, c, DEV, True, order, ns)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for k in ("out", "du", "ddelta", "dz", "dB", "dC", "dA", "dD", "ddelta_bias", "last_state"):
        a, b = fast[k].float(), slow[k].float()
        assert (a - b).abs().max() <= tol * max(1.0, float(b.abs().max())), k


# ---- volume -> tokens with LayerNorm: against F.layer_norm of the transposed tensor (fp32 reference) ------------------
@pytest.mark.parametrize("shape", [(2, 48, 4096), (2, 96, 1000), (1, 192, 512), (2, 384, 64), (1, 8, 24)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_tokens_matches_layer_norm(hip, shape, dtype):
    B, Cc, S = shape
    if dtype == torch.float32 and Cc > 192:
        pytest.skip("fp32 tiles are limited to 192 channels")
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = (1.5 * torch.randn(shape, device=DEV, generator=g) + 0.5).to(dtype)
    gamma, beta = torch.randn(Cc, device=DEV, generator=g), torch.randn(Cc, device=DEV, generator=g)
    dy = torch.randn(B, S, Cc, device=DEV, generator=g).to(dtype)
    xr = x.float().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr.transpose(1, 2), (Cc,), gr, br, 1e-5)
    gx, gg, gb = torch.autograd.grad(ref, (xr, gr, br), dy.float())
    y, mean, rstd = ops_raw.layernorm_tokens_fwd(hip, x, gamma, beta, 1e-5)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    dx, dgm, dbt = ops_raw.layernorm_tokens_bwd(hip, x, dy, mean, rstd, gamma)
    assert (dx.float() - gx).abs().max() <= tol * max(1.0, float(gx.abs().max()))
    assert (dgm - gg).abs().max() <= 2e-3 * max(1.0, float(gg.abs().max()))
    assert (dbt - gb).abs().max() <= 2e-3 * max(1.0, float(gb.abs().max()))
    d2 = ops_raw.layernorm_tokens_bwd(hip, x, dy, mean, rstd, gamma)
    assert torch.equal(d2[0], dx) and torch.equal(d2[1], dgm) and torch.equal(d2[2], dbt)      # deterministic


def test_conv3d_kernels_narrow_first_layer(hip):
    """4 -> 48 (the first SegMamba convolution): forward and weight gradient against fp32 torch."""
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(2, 4, 16, 16, 64, device=DEV, generator=g).bfloat16()
    w = (0.2 * torch.randn(48, 4, 3, 3, 3, device=DEV, generator=g)).bfloat16()
    dy = torch.randn(2, 48, 16, 16, 64, device=DEV, generator=g).bfloat16()
    ref = torch.nn.functional.conv3d(x.float(), w.float(), None, 1, 1)
    y = ops_raw.conv3d_k3_fwd(hip, x, ops_raw.pack_conv3d_weight(w))
    assert (y.float() - ref).abs().max() <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    dw = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
    ref_dw = _aten_wgrad(x.float(), dy.float())
    assert dw.shape == (48, 4, 3, 3, 3) and (dw - ref_dw).abs().max() <= 1e-4 * ref_dw.abs().max()


def test_conv3d_kernels_fp16(hip):
    """fp16 (the reference's autocast default) through the same kernels: forward, data gradient as forward, weight gradient."""
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(2, 48, 8, 16, 64, device=DEV, generator=g).half()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=DEV, generator=g)).half()
    dy = torch.randn(2, 48, 8, 16, 64, device=DEV, generator=g).half()
    xr, wr = x.float().requires_grad_(), w.float().requires_grad_()
    ref = torch.nn.functional.conv3d(xr, wr, None, 1, 1)
    gx, gw = torch.autograd.grad(ref, (xr, wr), dy.float())
    y = ops_raw.conv3d_k3_fwd(hip, x, ops_raw.pack_conv3d_weight(w, torch.float16))
    assert (y.float() - ref).abs().max() <= 2.0 ** -10 * max(1.0, float(ref.abs().max()))
    dx = ops_raw.conv3d_k3_fwd(hip, dy, ops_raw.pack_conv3d_weight_for_dgrad(w, torch.float16))
    assert (dx.float() - gx).abs().max() <= 2.0 ** -10 * max(1.0, float(gx.abs().max()))
    dw = ops_raw.conv3d_k3_wgrad(hip, x, dy, torch.float32)
    assert (dw - gw).abs().max() <= 1e-4 * float(gw.abs().max())


# ---- training-step glue: clip + SGD over a tensor list, cross entropy with its gradient ----------------------------------
@pytest.mark.parametrize("max_norm", [2.0, 0.0])
def test_sgd_clip_step_matches_torch(hip, max_norm):
    g = torch.Generator(device=DEV).manual_seed(11)
    sizes = [1, 7, 16384, 16385, 1 << 20, 48 * 48 * 27, 768] + [5 + 3 * i for i in range(200)]
    flat = torch.randn(sum(sizes) + 1, device=DEV, generator=g)
    params, off = [], 1
    for n in sizes:                                       # views at odd element offsets: the unaligned (scalar) path too
        params.append(flat[off:off + n])
        off += n
    ref_p = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.SGD(ref_p, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True)
    mine = [p.clone() for p in params]
    mom = [torch.zeros_like(p) for p in mine]
    for step in range(3):
        grads = [torch.randn(n, device=DEV, generator=g) for n in sizes]
        for p, gr in zip(ref_p, grads):
            p.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref_p, max_norm) if max_norm > 0 else None
        opt.step()
        head = ops_raw.sgd_clip_step(hip, mine, grads, mom, 1e-2, 0.99, 3e-5, True, max_norm)
        if norm is not None:
            assert abs(float(head[1]) - float(norm)) <= 1e-5 * float(norm)
    for a, b in zip(mine, ref_p):
        assert (a - b.detach()).abs().max() <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("shape,dtype", [((2, 4, 32, 32, 32), torch.bfloat16), ((1, 13, 1000), torch.float32), ((3, 2, 77), torch.float16)])
def test_cross_entropy_matches_torch(hip, shape, dtype):
    from segmamba_amd import train_ops
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    logits = (3 * torch.randn(shape, device=DEV, generator=g)).to(dtype).requires_grad_()
    labels = torch.randint(0, shape[1], (shape[0],) + shape[2:], device=DEV, generator=g)
    labels.view(-1)[::11] = -100
    ref_in = logits.detach().double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, labels)
    ref.backward()
    loss = train_ops.cross_entropy(logits, labels)
    loss.backward()
    assert loss.dtype == torch.float32 and abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    scale = float(ref_in.grad.abs().max())
    tol = 1e-6 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1e-2)
    assert (logits.grad.double() - ref_in.grad).abs().max() <= tol * scale


@pytest.mark.parametrize("B,Cin,Cout,S,dtype,bias", [(2, 48, 48, 64 ** 3, torch.bfloat16, True), (2, 4, 48, 64 ** 3, torch.bfloat16, False),
                                                   (2, 48, 4, 32 ** 3, torch.float16, True), (1, 96, 48, 32 ** 3, torch.bfloat16, True),
                                                   (2, 96, 96, 16 ** 3 * 8, torch.bfloat16, False)])
def test_pointwise_cf_matches_matmul(B, Cin, Cout, S, dtype, bias):
    """segm_pointwise_cf (1x1x1 convolution on channel-first activations) against an fp32 matmul of the same 16-bit operands, a
    channel-slice view as input, accumulation; and the autograd route of linear.pointwise against the BLAS route"""
    from segmamba_amd import linear as LN
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(Cin * 100 + Cout)
    full = torch.randn(B, Cin + 16, S, device=DEV, generator=g).to(dtype)
    x = full[:, 8:8 + Cin]
    w = (0.2 * torch.randn(Cout, Cin, device=DEV, generator=g)).to(dtype)
    b = torch.randn(Cout, device=DEV, generator=g) if bias else None
    y = ops_raw.pointwise_cf(hip, x, w, b)
    ref = torch.einsum("oc,bcs->bos", w.float(), x.float()) + (b.view(1, -1, 1) if bias else 0)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    y2 = ops_raw.pointwise_cf(hip, x, w, None, out=y.clone(), accumulate=True)
    ref2 = y.float() + torch.einsum("oc,bcs->bos", w.float(), x.float())
    assert (y2.float() - ref2).abs().max() <= tol * max(1.0, float(ref2.abs().max()))
    if S % 64 == 0 and Cin >= 8:
        res = []
        dy = torch.randn(B, Cout, S, device=DEV, generator=g).to(dtype)
        for on in (False, True):
            old = LN._PW_HIP
            LN._PW_HIP = on
            try:
                xa, wa = x.clone().requires_grad_(), w.clone().requires_grad_()
                ba = b.to(dtype).clone().requires_grad_() if bias else None
                out = LN.pointwise(xa.view(B, Cin, S, 1, 1), wa, ba).view(B, Cout, S)
                out.backward(dy)
                res.append([t.float() for t in (out.detach(), xa.grad, wa.grad)])
            finally:
                LN._PW_HIP = old
        for a, c in zip(*res):
            assert (a - c).abs().max() <= 3e-2 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,size,dtype", [(2, 4, 48, 128, torch.bfloat16), (1, 1, 48, 64, torch.float16), (1, 4, 32, 32, torch.bfloat16)])
def test_stem_conv_matches_conv3d(B, Cin, Cout, size, dtype):
    """segm_stem_conv_fwd (7^3 stride 2 padding 3) against conv3d in fp32 on the same 16-bit operands, at the BASELINE input size;
    and fused_norm.stem_conv3d's autograd (weight gradient through segm_stem_conv_wgrad for W in {64, 128, 256}, ATen otherwise) against F.conv3d's"""
    from segmamba_amd import fused_norm as FN
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(size + Cin)
    x = torch.rand(B, Cin, size, size, size, device=DEV, generator=g).to(dtype)
    w = (0.05 * torch.randn(Cout, Cin, 7, 7, 7, device=DEV, generator=g)).to(dtype)
    b = torch.randn(Cout, device=DEV, generator=g)
    y = ops_raw.stem_conv_fwd(hip, x, w, b)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), b, stride=2, padding=3)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    wa, ba = w.clone().requires_grad_(), b.to(dtype).clone().requires_grad_()
    wb, bb = w.clone().requires_grad_(), b.to(dtype).clone().requires_grad_()
    dy = torch.randn_like(y)
    FN.stem_conv3d(x, wa, ba).backward(dy)
    torch.nn.functional.conv3d(x, wb, bb, stride=2, padding=3).backward(dy)
    assert (wa.grad.float() - wb.grad.float()).abs().max() <= 3e-2 * max(1.0, float(wb.grad.float().abs().max()))
    assert (ba.grad.float() - bb.grad.float()).abs().max() <= 3e-2 * max(1.0, float(bb.grad.float().abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,size,dtype", [(2, 4, 48, 128, torch.bfloat16), (1, 3, 40, 64, torch.float16)])
def test_stem_conv_wgrad_matches_autograd(B, Cin, Cout, size, dtype):
    """segm_stem_conv_wgrad at the BASELINE input size against the fp32 weight gradient of conv3d on the same 16-bit operands
    (fp32 accumulation over 524 288 voxels: 1e-3 of the largest entry), and bitwise repeatable"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(size + Cout)
    x = torch.rand(B, Cin, size, size, size, device=DEV, generator=g).to(dtype)
    dy = torch.randn(B, Cout, size // 2, size // 2, size // 2, device=DEV, generator=g).to(dtype)
    x4 = ops_raw.stem_channel_last4(x)
    dw = ops_raw.stem_conv_wgrad(hip, x4, dy, Cin)
    w = torch.zeros(Cout, Cin, 7, 7, 7, device=DEV, requires_grad=True)
    torch.nn.functional.conv3d(x.float(), w, None, stride=2, padding=3).backward(dy.float())
    assert (dw - w.grad).abs().max() <= 1e-3 * float(w.grad.abs().max())
    assert torch.equal(dw, ops_raw.stem_conv_wgrad(hip, x4, dy, Cin))


@pytest.mark.parametrize("K,M,N,lda,ldb", [(524288, 192, 48, 192, 48), (524288, 35, 96, 35, 96), (65536, 192, 6, 192, 38), (70001, 100, 130, 104, 136),
                                           (524288, 36, 96, 40, 96), (65535, 96, 38, 96, 40)])
def test_wgrad_gemm_tn_matches_fp32_product(K, M, N, lda, ldb):
    """segm_wgrad_gemm (TN) at the stage-0 / stage-1 token counts against the fp64-accumulated product of the same bf16 operands
    (K up to 524 288 terms per entry: 1e-3 of the largest entry), bitwise repeatable"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(K % 1000 + M)
    a = torch.randn(K, lda, device=DEV, generator=g).bfloat16()[:, :M]
    b = torch.randn(K, ldb, device=DEV, generator=g).bfloat16()[:, :N]
    out = ops_raw.wgrad_gemm(hip, a, b, ops_raw.WGEMM_TN)
    ref = (a.double().t() @ b.double()).float()
    assert (out - ref).abs().max() <= 1e-3 * float(ref.abs().max())
    assert torch.equal(out, ops_raw.wgrad_gemm(hip, a, b, ops_raw.WGEMM_TN))


@pytest.mark.parametrize("K,M,N,lda,ldb,dtype", [(524288, 96, 3, 96, 40, torch.bfloat16), (65536, 192, 6, 192, 40, torch.bfloat16),
                                                 (8192, 384, 12, 384, 48, torch.bfloat16), (4096, 768, 24, 768, 56, torch.bfloat16),
                                                 (524288, 96, 3, 96, 35, torch.float16)])
def test_skinny_tn_matches_fp64_product(K, M, N, lda, ldb, dtype):
    """segm_skinny_tn at the four stages' dt_proj weight-gradient shapes (reference selective_scan_interface.py:272: ddelta^T x_dbl[:, :R])
    against the fp64 product of the same operands; the skinny operand is a column window of the x_proj output; bitwise repeatable"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(K % 1000 + M)
    a = torch.randn(K, lda, device=DEV, generator=g).to(dtype)[:, :M]
    b = torch.randn(K, ldb, device=DEV, generator=g).to(dtype)[:, :N]
    assert ops_raw.skinny_tn_supported(a, b)
    out = ops_raw.skinny_tn(hip, a, b)
    ref = (a.double().t() @ b.double()).float()
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert (out - ref).abs().max() <= 1e-3 * float(ref.abs().max())
    assert torch.equal(out, ops_raw.skinny_tn(hip, a, b))


@pytest.mark.parametrize("order,ns", [(L.TIME_FORWARD, 1), (L.TIME_REVERSED, 1), (L.TIME_INTERLEAVED, 64)])
def test_scan_with_conv1d_inside_the_launch_equals_conv1d_then_scan(order, ns):
    """`conv_weight=` at the stage-0 size (B=2, D=96, L=64^3, bf16): the scan launches that form u = SiLU(conv1d(x) + b) themselves
    against segm_causal_conv1d_fwd followed by the scan - bit for bit (output, gated output, checkpoints)"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(11 + order)
    Bn, D, N, Lq = 2, 96, 16, 64 ** 3
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g).bfloat16()
    x, z = rn(Bn, Lq, D), rn(Bn, Lq, D)
    delta = (0.5 * torch.rand(Bn, Lq, D, device=DEV, generator=g)).bfloat16()
    A = -0.5 * torch.rand(D, N, device=DEV, generator=g)
    Bm, Cm = rn(Bn, Lq, N), rn(Bn, Lq, N)
    Dv, db = torch.randn(D, device=DEV, generator=g), 0.5 * torch.rand(D, device=DEV, generator=g)
    cw, cb = 0.5 * torch.randn(D, 4, device=DEV, generator=g), 0.1 * torch.randn(D, device=DEV, generator=g)
    u = ops_raw.conv1d_fwd(hip, x, cw, cb, True, channel_last=True, time_order=order, nslices=ns)
    kw = dict(channel_last=True, time_order=order, nslices=ns, need_out=True, need_ckpt=True)
    ref = ops_raw.scan_fwd(hip, u, delta, A, Bm, Cm, Dv, z, db, True, **kw)
    assert ops_raw.scan_fused_conv_supported(hip, Bn, D, Lq, ns, order)
    fused = ops_raw.scan_fwd(hip, x, delta, A, Bm, Cm, Dv, z, db, True, conv_weight=cw, conv_bias=cb, **kw)
    for k in ("out", "out_z", "ckpt"):
        assert torch.equal(fused[k], ref[k]), k


@pytest.mark.parametrize("shape,dtype,padded", [((2, 48, 64, 64, 64), torch.bfloat16, False), ((2, 96, 262144), torch.bfloat16, False),
                                                ((2, 48, 128, 128, 128), torch.bfloat16, True), ((2, 4, 128, 128, 128), torch.float16, False),
                                                ((2, 384, 8, 8, 8), torch.float32, False)])
def test_channel_sum_matches_fp64_sum(shape, dtype, padded):
    """segm_channel_sum at the network's bias-gradient shapes (dense and padded-volume rows) against the fp64 sum of the same
    values; bitwise repeatable"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    if padded:
        x = ops_raw.volume_empty(shape[0], shape[1], shape[2:], dtype, DEV)
        x.copy_(torch.randn(shape, device=DEV, generator=g))
        assert not x.is_contiguous()
    else:
        x = torch.randn(shape, device=DEV, generator=g).to(dtype)
    out = ops_raw.channel_sum(hip, x)
    dims = [0] + list(range(2, x.dim()))
    ref = x.double().sum(dims)
    assert (out.double() - ref).abs().max() <= 1e-5 * float(x.double().abs().sum(dims).max())
    assert torch.equal(out, ops_raw.channel_sum(hip, x))


@pytest.mark.parametrize("Bn,M,N,K", [(2, 48, 48, 128 ** 3), (2, 4, 48, 128 ** 3), (2, 96, 96, 64 ** 3)])
def test_wgrad_gemm_nt_matches_fp32_product(Bn, M, N, K):
    """segm_wgrad_gemm (NT) on channel-first volumes of the BASELINE size (padded channel stride as the convolutions write them)"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(M + N)
    a = ops_raw.volume_empty(Bn, M, (K,), torch.bfloat16, DEV)
    a.copy_(torch.randn(Bn, M, K, device=DEV, generator=g))
    b = torch.randn(Bn, N, K, device=DEV, generator=g).bfloat16()
    out = ops_raw.wgrad_gemm(hip, a, b, ops_raw.WGEMM_NT)
    ref = torch.einsum("bmk,bnk->mn", a.double(), b.double()).float()
    assert (out - ref).abs().max() <= 1e-3 * float(ref.abs().max())
    assert torch.equal(out, ops_raw.wgrad_gemm(hip, a, b, ops_raw.WGEMM_NT))


def test_thin_input_conv3_matches_conv3d_at_size():
    """the thin-input kernels as the 3x3x3 stride-1 first layer (4 -> 48 at 2 x 128^3, the BASELINE input): forward against conv3d and
    weight gradient against autograd, fp32 on the same bf16 operands; the output comes with the padded channel stride of 128^3 volumes"""
    hip = L.get_lib()
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(2, 4, 128, 128, 128, device=DEV, generator=g).bfloat16()
    w = (0.1 * torch.randn(48, 4, 3, 3, 3, device=DEV, generator=g)).bfloat16()
    y = ops_raw.stem_conv_fwd(hip, x, w, None)
    assert not y.is_contiguous() and ops_raw.channel_dense(y)
    wr = w.float().requires_grad_()
    ref = torch.nn.functional.conv3d(x.float(), wr, None, stride=1, padding=1)
    assert (y.float() - ref).abs().max() <= 1e-2 * max(1.0, float(ref.abs().max()))
    dy = ops_raw.volume_empty(2, 48, (128, 128, 128), torch.bfloat16, DEV)
    dy.copy_(torch.randn(2, 48, 128, 128, 128, device=DEV, generator=g))
    ref.backward(dy.float())
    dw = ops_raw.stem_conv_wgrad(hip, ops_raw.stem_channel_last4(x), dy, 4, 3)
    assert (dw - wr.grad).abs().max() <= 1e-3 * float(wr.grad.abs().max())


@pytest.mark.parametrize("shape,dtype", [((2, 48, 64, 64, 64), torch.bfloat16), ((1, 96, 32, 32, 32), torch.float16), ((2, 4, 3, 5, 24), torch.bfloat16)])
def test_depth_to_space2(hip, shape, dtype):
    """Round 4: the permute behind a kernel-2 stride-2 transposed convolution (reference unetr_block.py:52-60) and its inverse as
    one library kernel each - bit for bit against ATen's reshape / permute, at the decoder's two large shapes, into dense volumes
    and into a volume with a padded channel stride (ops_raw.volume_empty)."""
    B, Cc, D, Hh, W = shape
    torch.manual_seed(sum(shape))
    blk = torch.randn(B, Cc * 8, D, Hh, W, device=DEV).to(dtype)
    ref = blk.reshape(B, Cc, 2, 2, 2, D, Hh, W).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, Cc, 2 * D, 2 * Hh, 2 * W)
    vol = ops_raw.depth_to_space2(hip, blk)
    assert torch.equal(vol, ref)
    padded = ops_raw.volume_empty(B, Cc, (2 * D, 2 * Hh, 2 * W), dtype, torch.device(DEV))
    ops_raw.depth_to_space2(hip, blk, out=padded)
    assert torch.equal(padded, ref)
    assert torch.equal(ops_raw.space_to_depth2(hip, padded), blk)
    assert torch.equal(ops_raw.space_to_depth2(hip, vol), blk)


# ---- channel-last 3x3x3 convolution (round 6 prototype, csrc/conv3d_cl.hip) ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("waves8", [False, True])
def test_conv3d_k3_channel_last_forward_and_dgrad_at_the_benchmarked_shape(waves8):
    """48 -> 48 at 2 x 128^3, channels last: forward (+ bias) and data gradient (the same kernel on dy with the image of
    flip(w).transpose(0, 1)) against fp32 ATen on the same bf16-rounded operands; the bound of the NCDHW at-size test (1e-2 of the
    largest reference value).  Also one 2 x 64^3 accumulate call on padded voxel strides."""
    from segmamba_amd import lib as L, ops_raw
    hip = L.get_lib()
    dev = torch.device("cuda")
    B, S = 2, 128
    g = torch.Generator(device=dev).manual_seed(5)
    x = (0.5 * torch.randn(B, 48, S, S, S, device=dev, generator=g)).bfloat16()
    w = (torch.randn(48, 48, 3, 3, 3, device=dev, generator=g) / (27 * 48) ** 0.5).bfloat16()
    bias = torch.randn(48, device=dev, generator=g)
    dy = (0.5 * torch.randn(B, 48, S, S, S, device=dev, generator=g)).bfloat16()
    xcl = x.permute(0, 2, 3, 4, 1).contiguous()
    y = ops_raw.conv3d_k3_fwd_cl(hip, xcl, ops_raw.conv3d_cl_weight_image(hip, w), bias, waves8=waves8)
    wt = torch.flip(w, (2, 3, 4)).transpose(0, 1).contiguous()
    dx = ops_raw.conv3d_k3_fwd_cl(hip, dy.permute(0, 2, 3, 4, 1).contiguous(), ops_raw.conv3d_cl_weight_image(hip, wt), None, waves8=waves8)
    for b in range(B):                                    # fp32 activations at this size are 1.6 GB per batch of two
        xr = x[b:b + 1].float().requires_grad_()
        yr = torch.nn.functional.conv3d(xr, w.float(), bias, 1, 1)
        yr.backward(dy[b:b + 1].float())
        sc = float(yr.detach().abs().max())
        assert float((y[b:b + 1].permute(0, 4, 1, 2, 3).float() - yr.detach()).abs().max()) <= 1e-2 * sc
        sc = float(xr.grad.abs().max())
        assert float((dx[b:b + 1].permute(0, 4, 1, 2, 3).float() - xr.grad).abs().max()) <= 1e-2 * sc
        del xr, yr
    S2 = 64
    x2 = torch.zeros(B, S2, S2, S2, 56, device=dev, dtype=torch.bfloat16)
    x2[..., :48] = 0.5 * torch.randn(B, S2, S2, S2, 48, device=dev, generator=g)
    out = torch.zeros(B, S2, S2, S2, 64, device=dev, dtype=torch.bfloat16)
    out[..., :48] = torch.randn(B, S2, S2, S2, 48, device=dev, generator=g)
    ref = out[..., :48].permute(0, 4, 1, 2, 3).float() + torch.nn.functional.conv3d(x2[..., :48].permute(0, 4, 1, 2, 3).float(), w.float(), None, 1, 1)
    ops_raw.conv3d_k3_fwd_cl(hip, x2[..., :48], ops_raw.conv3d_cl_weight_image(hip, w), None, out=out[..., :48], accumulate=True, waves8=waves8)
    assert float((out[..., :48].permute(0, 4, 1, 2, 3).float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
    assert bool((out[..., 48:] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_add3_on_the_gpu(dtype):
    """segm_add3 at the size of a stage-0 dxz tensor (2 x 262144 x 192) and at a small odd packet count: == fp32 sum rounded once"""
    from segmamba_amd import lib as L, ops_raw
    hip = L.get_lib()
    g = torch.Generator(device="cuda").manual_seed(4)
    for shape in ((2, 262144, 192), (3, 8 * 37)):
        a, b, c = (torch.randn(*shape, device="cuda", generator=g).to(dtype) for _ in range(3))
        ref = (a.float() + b.float() + c.float()).to(dtype)
        assert torch.equal(ops_raw.add3(hip, a, b, c), ref)
        assert torch.equal(ops_raw.add3(hip, a, b, c, out=a), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,S,dtype", [(384, 384, 16, torch.bfloat16), (768, 768, 8, torch.bfloat16), (384, 192, 32, torch.bfloat16),
                                              (192, 384, 16, torch.float16), (96, 192, 32, torch.bfloat16)])
def test_conv3d_k3_cube_at_the_benchmarked_shapes(cin, cout, S, dtype):
    """segm_conv3d_k3_cube_fwd (ABI 10) on the wide layers of the 8^3 / 16^3 / 32^3 levels, batch 2: forward (+ bias, + InstanceNorm
    partials) and data gradient (the same launch on dy with the flipped image) against fp32 ATen on the same 16-bit operands - the
    1e-2 of the largest reference value bound of the other convolution kernels' at-size tests; every plan (column tiles x splits) of
    one layer gives the same result within that bound; run-to-run bit-identical (fixed-order reduction, no atomics); a second
    input part accumulated in place"""
    from segmamba_amd import lib as L, ops_raw
    hip = L.get_lib()
    dev = torch.device("cuda")
    B = 2
    g = torch.Generator(device=dev).manual_seed(cin + cout + S)
    x = (0.5 * torch.randn(B, cin, S, S, S, device=dev, generator=g)).to(dtype)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) / (27 * cin) ** 0.5).to(dtype)
    bias = torch.randn(cout, device=dev, generator=g)
    dy = (0.5 * torch.randn(B, cout, S, S, S, device=dev, generator=g)).to(dtype)
    xr = x.float().requires_grad_()
    yr = torch.nn.functional.conv3d(xr, w.float(), bias, 1, 1)
    yr.backward(dy.float())
    yr = yr.detach()
    img, imgT = ops_raw.conv3d_cube_weight_image(hip, w), ops_raw.conv3d_cube_weight_image(hip, w, flipped=True)
    # the pack kernel (what the device path uses) against the gather through the exported index map, also on a channel slice
    assert torch.equal(img, ops_raw.conv3d_cube_weight_image(hip, w, by_index=True))
    assert torch.equal(imgT, ops_raw.conv3d_cube_weight_image(hip, w, flipped=True, by_index=True))
    if cin >= 64:
        part = w[:, 32:]
        for fl in (False, True):
            assert torch.equal(ops_raw.conv3d_cube_weight_image(hip, part, fl), ops_raw.conv3d_cube_weight_image(hip, part.contiguous(), fl, by_index=True))
    y, st = ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias, want_stats=True)
    assert float((y.float() - yr).abs().max()) <= 1e-2 * float(yr.abs().max())
    assert torch.equal(y, ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias))
    dx = ops_raw.conv3d_k3_cube_fwd(hip, dy, imgT, cin)
    assert float((dx.float() - xr.grad).abs().max()) <= 1e-2 * float(xr.grad.abs().max())
    # statistics partials -> the same mean / rstd as InstanceNorm's own pass over y
    _, m1, r1 = ops_raw.instnorm_fwd(hip, y, None, "none", stats=st)
    _, m0, r0 = ops_raw.instnorm_fwd(hip, y, None, "none")
    assert float((m1 - m0).abs().max()) <= 1e-2 * max(1.0, float(m0.abs().max())) and float((r1 / r0 - 1).abs().max()) <= 1e-2
    R = cin // 32
    for nt in (2, 3, 4):
        if cout % (32 * nt):
            continue
        for s in sorted({1, R, max(d for d in range(1, R + 1) if R % d == 0 and d <= 4)}):
            yy = ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, bias, nt=nt, splits=s)
            assert float((yy.float() - yr).abs().max()) <= 1e-2 * float(yr.abs().max()), (nt, s)
    out = y.clone()
    ops_raw.conv3d_k3_cube_fwd(hip, x, img, cout, None, out=out, accumulate=True)
    assert float((out.float() - (2 * yr - bias.view(1, -1, 1, 1, 1))).abs().max()) <= 2e-2 * float(yr.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,S,dtype", [(384, 384, 8, torch.bfloat16), (768, 768, 8, torch.bfloat16), (384, 768, 8, torch.float16),
                                              (384, 384, 16, torch.bfloat16), (192, 192, 32, torch.bfloat16)])
def test_conv3d_k3_cube_wgrad_at_the_benchmarked_shapes(cin, cout, S, dtype):
    """segm_conv3d_k3_cube_wgrad, batch 2, against fp32 ATen's weight gradient on the same 16-bit operands: fp32 result within 1e-3
    of the largest reference value (fp32 accumulation of exact products, another order), 16-bit result within the 1e-2 bound of the
    other kernels; run-to-run bit-identical; and through the dispatcher (conv3d._wgrad) where the routing table picks it"""
    from segmamba_amd import conv3d as C, lib as L, ops_raw
    hip = L.get_lib()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(cin + cout + S)
    x = (0.5 * torch.randn(2, cin, S, S, S, device=dev, generator=g)).to(dtype)
    dy = (0.5 * torch.randn(2, cout, S, S, S, device=dev, generator=g)).to(dtype)
    w = torch.zeros(cout, cin, 3, 3, 3, device=dev)
    ref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, [1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1, [False, True, False])[1]
    sc = float(ref.abs().max())
    dw = ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, torch.float32)
    assert float((dw - ref).abs().max()) <= 1e-3 * sc
    assert torch.equal(dw, ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, torch.float32))
    assert float((ops_raw.conv3d_k3_cube_wgrad(hip, x, dy, dtype).float() - ref).abs().max()) <= 1e-2 * sc
    if S == 8 or (S == 16 and cin * cout >= 384 * 384):    # the shapes the routing table gives to this kernel
        assert C._cube_wgrad_ok(x, dy, w)
        got = C._wgrad(x, dy, w.to(dtype), 1, torch.float32)
        assert got.dtype == torch.float32 and float((got - ref).abs().max()) <= 1e-3 * sc
