"""Golden fixtures for the out-projection variants of the fused inner function, from the REFERENCE's own code.

Run in the build container only (needs /root/reference, CPU):
    python tests/golden/make_golden_inner_out_proj.py

  inner_fn_vB{0,1}_vC{0,1}.npz   reference `mamba_inner_ref` (mamba/mamba_ssm/ops/selective_scan_interface.py:636-670) on the
                                 reference's own `selective_scan_ref` / `causal_conv1d_ref`, with the distributions of the
                                 reference test (mamba/tests/ops/test_selective_scan.py:152-221: dstate 8, conv width 3, real A,
                                 B / C either input-dependent (None) or constant (dim, dstate)) at a reduced dim (64, dt_rank 4;
                                 the reference test uses 768 / 48 - the GPU test runs that size against the pinned oracle).
  bimamba_inner.npz              reference `bimamba_inner_ref` (:673-709), variable B / C.
Outputs and every gradient are stored.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden import load_reference, save  # noqa: E402


def make_inputs(is_variable_B, is_variable_C, dim=64, dstate=8, dt_rank=4, seqlen=128, batch=2, seed=0):
    """test_selective_scan.py:169-190, fp32, real A"""
    torch.random.manual_seed(seed)
    t = {}
    t["xz"] = torch.randn(batch, 2 * dim, seqlen, requires_grad=True)
    t["conv_w"] = torch.randn(dim, 1, 3, requires_grad=True)
    t["conv_b"] = torch.randn(dim, requires_grad=True)
    t["x_proj_w"] = torch.randn(dt_rank + (bool(is_variable_B) + bool(is_variable_C)) * dstate, dim, requires_grad=True)
    t["dt_proj_w"] = torch.randn(dim, dt_rank, requires_grad=True)
    t["out_proj_w"] = torch.randn(dim // 2, dim, requires_grad=True)
    t["A"] = (-0.5 * torch.rand(dim, dstate)).requires_grad_()
    t["B"] = torch.randn(dim, dstate, requires_grad=True) if not is_variable_B else None
    t["C"] = torch.randn(dim, dstate, requires_grad=True) if not is_variable_C else None
    t["D"] = torch.randn(dim, requires_grad=True)
    t["delta_bias"] = (0.5 * torch.rand(dim)).requires_grad_()
    return t


def main():
    cci, ssi, ms = load_reference()
    for vB in (False, True):
        for vC in (False, True):
            t = make_inputs(vB, vC)
            out = ssi.mamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                                      t["A"], t["B"], t["C"], t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
            g = torch.randn_like(out)
            out.backward(g)
            save(f"inner_fn_vB{int(vB)}_vC{int(vC)}.npz", g=g, out=out,
                 **{k: v for k, v in t.items() if v is not None},
                 **{"d" + k: v.grad for k, v in t.items() if v is not None})
    t = make_inputs(True, True, seed=1)
    t["A_b"] = (-0.5 * torch.rand(64, 8)).requires_grad_()
    out = ssi.bimamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                                t["A"], t["A_b"], None, None, t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    g = torch.randn_like(out)
    out.backward(g)
    save("bimamba_inner.npz", g=g, out=out, **{k: v for k, v in t.items() if v is not None},
         **{"d" + k: v.grad for k, v in t.items() if v is not None})


if __name__ == "__main__":
    main()
