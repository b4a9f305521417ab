"""GPU parity tests of the native code written after round 1's GPU budget was spent: parity-tested on the CPU emulation of HIP
(tests/test_emu_kernels.py) but never yet run on an MI355X.  They live in this file - collected last - so that a surprise
here cannot hide the results of the tests before it."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from segmamba_amd import lib as L
from segmamba_amd import ops_raw

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    return L.get_lib()


@pytest.mark.parametrize("shape", [(1, 48, 8, 8, 32), (2, 48, 16, 16, 128), (1, 96, 6, 16, 16), (2, 48, 2, 20, 8), (1, 48, 3, 40, 72)])
def test_conv3d_k3_fwd_newer_chained_variants(hip, shape):
    """SEGM_CONV_FWD_PITCH48 (unpadded LDS rows) and SEGM_CONV_FWD_CHAIN32 (32-wide x blocks, two workgroups per CU): the same
    sums in the same order as the chained kernel, so bit-identical results; plus in-place accumulation."""
    B, cout, D, H_, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape) + 5)
    x = torch.randn(B, 96, D, H_, W, device=DEV, generator=g).bfloat16()
    w = (0.05 * torch.randn(cout, 96, 3, 3, 3, device=DEV, generator=g)).bfloat16()
    bias = torch.randn(cout, device=DEV, generator=g)
    w0, w1 = ops_raw.pack_conv3d_weight(w[:, :48]), ops_raw.pack_conv3d_weight(w[:, 48:])
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias, 1, 1)
    tol = 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    y = ops_raw.conv3d_k3_fwd(hip, x[:, :48], w0, bias, chain=True)
    for kw in (dict(chain=True, pitch48=True), dict(chain32=True)):
        assert torch.equal(ops_raw.conv3d_k3_fwd(hip, x[:, :48], w0, bias, **kw), y), kw
        acc = y.clone()
        ops_raw.conv3d_k3_fwd(hip, x[:, 48:], w1, None, out=acc, accumulate=True, **kw)
        assert (acc.float() - ref).abs().max() <= 2 * tol, kw


@pytest.mark.parametrize("idx", [-1, -2])
def test_conv3d_same_autograd_with_newer_library_candidates(hip, monkeypatch, idx):
    """the dispatcher with its last (chained, 32-wide) / second-to-last (chained, unpadded rows) forward candidates forced in"""
    from segmamba_amd import conv3d as C3
    monkeypatch.setenv("SEGM_CONV_FWD_UNTIMED", "1")
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(2, 48, 8, 16, 32, device=DEV, generator=g).bfloat16().requires_grad_()
    w = (0.05 * torch.randn(48, 48, 3, 3, 3, device=DEV, generator=g)).bfloat16().requires_grad_()
    bias = torch.randn(48, device=DEV, generator=g).bfloat16().requires_grad_()
    dy = torch.randn(2, 48, 8, 16, 32, device=DEV, generator=g).bfloat16()
    monkeypatch.setattr(C3, "_pick", lambda key, cands, *rest: cands[idx if key[0] != "wgrad" else -1]())
    y = C3.conv3d_same(x, w, bias)
    gx, gw, gb = torch.autograd.grad(y, (x, w, bias), dy)
    x2, w2, b2 = (t.detach().float().requires_grad_() for t in (x, w, bias))
    y2 = torch.nn.functional.conv3d(x2, w2, b2, 1, 1)
    gx2, gw2, gb2 = torch.autograd.grad(y2, (x2, w2, b2), dy.float())
    for got, want in ((y, y2), (gx, gx2), (gw, gw2), (gb, gb2)):
        assert (got.float() - want).abs().max() <= 2e-2 * max(1.0, float(want.abs().max()))


# ---- decode path -------------------------------------------------------------------------------------
def test_decode_step_kernels_match_reference_fixture(hip):
    f = {k: v.to(DEV) for k, v in H.load_golden("mamba_decode.npz").items() if torch.is_tensor(v) and v.dim() > 0}
    cs = f["cu.state_in"].clone()
    out = ops_raw.conv1d_update(hip, f["cu.x"], cs, f["cu.weight"], f["cu.bias"], True)
    H.assert_close(out, f["cu.out"], 1e-5, 1e-5, "conv update out")
    assert torch.equal(cs, f["cu.state_out"])
    st = f["su.state_in"].clone()
    out = ops_raw.state_update(hip, st, f["su.x"], f["su.dt"], f["su.A"], f["su.B"], f["su.C"], f["su.D"], f["su.z"], f["su.dt_bias"], True)
    H.assert_close(out, f["su.out"], 1e-4, 1e-5, "state update out")
    H.assert_close(st, f["su.state_out"], 1e-4, 1e-5, "state update state")
    # bf16 activations, fp32 state, strided x (the halves of xz)
    xz = torch.randn(3, 20, device=DEV).bfloat16()
    st = f["su.state_in"].clone()
    from oracle import ref_ops
    ref_st = st.clone()
    x, z = xz[:, :10], xz[:, 10:]
    dt = f["su.dt"].bfloat16()
    want = ref_ops.selective_state_update_ref(ref_st, x.float(), dt.float(), f["su.A"], f["su.B"].bfloat16().float(),
                                              f["su.C"].bfloat16().float(), f["su.D"], z=z.float(), dt_bias=f["su.dt_bias"], dt_softplus=True)
    got = ops_raw.state_update(hip, st, x, dt, f["su.A"], f["su.B"].bfloat16(), f["su.C"].bfloat16(), f["su.D"], z, f["su.dt_bias"], True)
    H.assert_close(got.float(), want, 2e-2, 2e-2, "bf16 state update out")
    H.assert_close(st, ref_st, 1e-4, 1e-4, "bf16 state update state")


def test_mamba_prefill_and_step_match_reference_fixture():
    """Mamba.forward(h, inference_params): prefill, then one token at a time, against the reference Mamba's own run."""
    import types
    from mamba_ssm import Mamba
    f = H.load_golden("mamba_decode.npz")
    m = Mamba(d_model=12, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=4, layer_idx=0)
    m.load_state_dict({k[len("param."):]: v for k, v in f.items() if k.startswith("param.")})
    m = m.to(DEV)
    h, L0 = f["h"].to(DEV), int(f["L0"])
    params = types.SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0)
    with torch.no_grad():
        out = m(h[:, :L0], inference_params=params)
        conv, ssm = params.key_value_memory_dict[0]
        H.assert_close(out, f["out_prefill"], 1e-3, 1e-4, "prefill out")
        H.assert_close(ssm, f["ssm_state_prefill"], 1e-3, 1e-4, "prefill ssm_state")
        outs = []
        for t in range(L0, h.shape[1]):
            params.seqlen_offset = t
            outs.append(m(h[:, t:t + 1], inference_params=params))
    H.assert_close(torch.cat(outs, 1), f["out_steps"], 1e-3, 1e-4, "step outs")
    H.assert_close(conv, f["conv_state_final"], 1e-4, 1e-5, "final conv_state")
    H.assert_close(ssm, f["ssm_state_final"], 1e-3, 1e-4, "final ssm_state")


@pytest.mark.parametrize("M,K,N", [(65536, 48, 192), (40000, 96, 48), (33000, 192, 96), (70001, 96, 384), (32768, 48, 36)])
def test_linear_rows_matches_fp32_linear(hip, M, K, N):
    """segm_linear_rows (opt-in route of linear.linear_cl, not yet timed) against F.linear in fp32 on the same operands."""
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    xw = torch.randn(M, 2 * K, device=DEV, generator=g).bfloat16()
    x = xw[:, K:]                                            # a column slice, as the x / z halves of xz are
    w = (0.1 * torch.randn(N, K, device=DEV, generator=g)).bfloat16()
    b = torch.randn(N, device=DEV, generator=g)
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    y = ops_raw.linear_rows(hip, x, w, b)
    assert (y.float() - ref).abs().max() <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    assert torch.equal(ops_raw.linear_rows(hip, x, w, b), y)
