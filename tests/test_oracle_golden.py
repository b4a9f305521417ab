"""Pin the CPU oracle (oracle/ref_ops.py) to fixtures produced by the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops

from tests.golden.make_golden import named_fill  # deterministic weights shared with the generator


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _close(a, b, tol=2e-5):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("L", [64, 256])
@pytest.mark.parametrize("G", [1, 2])
def test_scan_oracle_matches_reference(golden_dir, L, G):
    f = _load(golden_dir, f"scan_L{L}_G{G}.npz")
    leaves = {k: f[k].clone().requires_grad_() for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}
    out, last = ref_ops.selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"],
                                           leaves["D"], z=leaves["z"], delta_bias=leaves["delta_bias"],
                                           delta_softplus=True, return_last_state=True)
    _close(out, f["out"])
    _close(last, f["last_state"])
    out.backward(f["g"])
    for k, gk in (("u", "du"), ("delta", "ddelta"), ("A", "dA"), ("B", "dB"), ("C", "dC"), ("D", "dD"),
                  ("z", "dz"), ("delta_bias", "ddelta_bias")):
        _close(leaves[k].grad, f[gk], tol=1e-4)


@pytest.mark.parametrize("L", [64, 256])
def test_closed_form_backward_matches_reference_autograd(golden_dir, L):
    f = _load(golden_dir, f"scan_L{L}_G1.npz")
    g = ref_ops.selective_scan_bwd_closed_form(f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], f["z"],
                                               f["delta_bias"], f["g"], delta_softplus=True)
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"):
        _close(g[k].float(), f[k], tol=1e-4)


@pytest.mark.parametrize("width", [2, 3, 4])
def test_conv1d_oracle_matches_reference(golden_dir, width):
    f = _load(golden_dir, f"conv1d_w{width}.npz")
    x, w, b = (f[k].clone().requires_grad_() for k in ("x", "weight", "bias"))
    out = ref_ops.causal_conv1d_ref(x, w, b, activation="silu")
    _close(out, f["out"])
    out.backward(f["g"])
    _close(x.grad, f["dx"])
    _close(w.grad, f["dweight"], tol=1e-4)
    _close(b.grad, f["dbias"], tol=1e-4)
    _close(ref_ops.causal_conv1d_ref(f["x"], f["weight"], None, None), f["out_nobias_noact"])


def test_inner_no_out_proj_oracle_matches_reference(golden_dir):
    f = _load(golden_dir, "inner_no_out_proj.npz")
    names = ("xz", "conv_w", "conv_b", "x_proj_w", "dt_proj_w", "A", "D", "delta_bias")
    t = {k: f[k].clone().requires_grad_() for k in names}
    out = ref_ops.mamba_inner_no_out_proj_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"],
                                              t["A"], t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    _close(out, f["out"])
    out.backward(f["g"])
    for k, gk in (("xz", "dxz"), ("conv_w", "dconv_w"), ("conv_b", "dconv_b"), ("x_proj_w", "dx_proj_w"),
                  ("dt_proj_w", "ddt_proj_w"), ("A", "dA"), ("D", "dD"), ("delta_bias", "ddelta_bias")):
        _close(t[k].grad, f[gk], tol=2e-4)


_INNER = ("xz", "conv_w", "conv_b", "x_proj_w", "dt_proj_w", "out_proj_w", "A", "B", "C", "D", "delta_bias", "A_b")


@pytest.mark.parametrize("vB", [0, 1])
@pytest.mark.parametrize("vC", [0, 1])
def test_inner_fn_with_out_proj_oracle_matches_reference(golden_dir, vB, vC):
    """`mamba_inner_ref` with the output projection, B / C input-dependent or constant - the reference test's matrix
    (mamba/tests/ops/test_selective_scan.py:152-221) - pinned by tests/golden/make_golden_inner_out_proj.py"""
    f = _load(golden_dir, f"inner_fn_vB{vB}_vC{vC}.npz")
    t = {k: f[k].clone().requires_grad_() for k in _INNER if k in f}
    out = ref_ops.mamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                                  t["A"], t.get("B"), t.get("C"), t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    _close(out, f["out"], tol=1e-4)
    out.backward(f["g"])
    for k in t:
        _close(t[k].grad, f["d" + k], tol=5e-4)


def test_bimamba_inner_oracle_matches_reference(golden_dir):
    f = _load(golden_dir, "bimamba_inner.npz")
    t = {k: f[k].clone().requires_grad_() for k in _INNER if k in f}
    out = ref_ops.bimamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                                    t["A"], t["A_b"], None, None, t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    _close(out, f["out"], tol=1e-4)
    out.backward(f["g"])
    for k in t:
        _close(t[k].grad, f["d" + k], tol=5e-4)


def test_mamba_v3_oracle_matches_reference(golden_dir):
    f = _load(golden_dir, "mamba_v3.npz")
    ns = int(f["nslices"])
    shapes = {k[len("grad__"):]: v.shape for k, v in f.items() if k.startswith("grad__")}
    params = named_fill({k: torch.empty(s) for k, s in shapes.items()})
    params = {k: v.requires_grad_() for k, v in params.items()}
    x = f["x"].clone().requires_grad_()
    y = ref_ops.mamba_v3_forward_ref(x, params, ns)
    _close(y, f["y"])
    y.backward(f["g"])
    _close(x.grad, f["dx"], tol=2e-4)
    for k, p in params.items():
        _close(p.grad, f["grad__" + k], tol=5e-4)


def test_slice_interleave_roundtrip():
    x = torch.arange(2 * 3 * 24, dtype=torch.float32).reshape(2, 3, 24)
    for ns in (1, 2, 4, 8, 24):
        y = ref_ops.slice_interleave(x, ns)
        # reference: mamba_simple.py:245-247 -> chunk, stack(dim=-1), flatten(-2)
        ref = torch.stack(x.chunk(ns, dim=-1), dim=-1).flatten(-2)
        assert torch.equal(y, ref)
        assert torch.equal(ref_ops.slice_deinterleave(y, ns), x)


def test_decode_oracle_matches_reference(golden_dir):
    """the decode restatements (update ops, Mamba.step, prefill with inference_params) against the reference's own run."""
    f = _load(golden_dir, "mamba_decode.npz")
    cs = f["cu.state_in"].clone()
    _close(ref_ops.causal_conv1d_update_ref(f["cu.x"], cs, f["cu.weight"], f["cu.bias"], "silu"), f["cu.out"])
    _close(cs, f["cu.state_out"])
    st = f["su.state_in"].clone()
    out = ref_ops.selective_state_update_ref(st, f["su.x"], f["su.dt"], f["su.A"], f["su.B"], f["su.C"], f["su.D"], z=f["su.z"],
                                             dt_bias=f["su.dt_bias"], dt_softplus=True)
    _close(out, f["su.out"])
    _close(st, f["su.state_out"])
    p = {k[len("param."):]: v for k, v in f.items() if k.startswith("param.")}
    h, L0 = f["h"], int(f["L0"])
    conv = torch.zeros_like(f["conv_state_final"])
    ssm = torch.zeros_like(f["ssm_state_final"])
    _close(ref_ops.mamba_prefill_ref(h[:, :L0], p, conv, ssm), f["out_prefill"])
    _close(conv, f["conv_state_prefill"])
    _close(ssm, f["ssm_state_prefill"])
    outs = [ref_ops.mamba_step_ref(h[:, t:t + 1], p, conv, ssm) for t in range(L0, h.shape[1])]
    _close(torch.cat(outs, 1), f["out_steps"])
    _close(conv, f["conv_state_final"])
    _close(ssm, f["ssm_state_final"])


# ---- the fp64 C oracle (oracle/scan_ref.c): the checker of the at-size GPU tests, pinned to the same fixtures --------------
@pytest.mark.parametrize("L", [64, 256])
@pytest.mark.parametrize("G", [1, 2])
def test_c_oracle_matches_reference(golden_dir, L, G):
    from oracle import scan_ref
    f = _load(golden_dir, f"scan_L{L}_G{G}.npz")
    args = (f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], f["z"], f["delta_bias"])
    r = scan_ref.scan_fwd(*args, delta_softplus=True)
    _close(r["out"].float(), f["out"])
    _close(r["last_state"].float(), f["last_state"])
    g = scan_ref.scan_bwd(*args, f["g"], delta_softplus=True)
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"):
        _close(g[k].float(), f[k], tol=2e-5)


def test_c_oracle_matches_pytorch_oracle_on_ragged_optional_cases():
    """lengths that are not multiples of the C oracle's recompute block, no z / D / bias, no softplus"""
    from oracle import scan_ref
    from tests import helpers as H
    for seqlen, has_z, has_D, has_bias, softplus in [(2049, True, True, True, True), (37, False, True, False, False),
                                                      (4100, True, False, True, True)]:
        c = H.scan_case(1, 6, 5, seqlen, seed=seqlen, has_z=has_z, has_D=has_D, has_bias=has_bias)
        ref = H.scan_oracle(c, softplus=softplus)
        args = (c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"])
        r = scan_ref.scan_fwd(*args, delta_softplus=softplus)
        _close(r["out"].float(), ref["out"])
        g = scan_ref.scan_bwd(*args, c["g"], delta_softplus=softplus)
        for k in g:
            _close(g[k].float(), ref[k], tol=1e-4)


@pytest.mark.parametrize("width", [2, 3, 4])
def test_c_oracle_conv1d_matches_reference(golden_dir, width):
    from oracle import scan_ref
    f = _load(golden_dir, f"conv1d_w{width}.npz")
    _close(scan_ref.conv1d_fwd(f["x"], f["weight"], f["bias"], True).float(), f["out"])
    _close(scan_ref.conv1d_fwd(f["x"], f["weight"], None, False).float(), f["out_nobias_noact"])
