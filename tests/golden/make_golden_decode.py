"""Golden fixture of the decode path, generated from the REFERENCE's own code (build container only: needs /root/reference).

    python tests/golden/make_golden_decode.py

mamba_decode.npz: reference `Mamba(bimamba_type="v3", layer_idx=0)` on CPU - `forward(h[:, :L0], inference_params)` at
seqlen_offset 0 (prefill: the reference's uni-directional branch, mamba_simple.py:265-355, with `causal_conv1d_fn` /
`selective_scan_fn` bound to the reference's own `*_ref` functions), then `forward` token by token at seqlen_offset > 0
(`Mamba.step`, :356-401, with `causal_conv1d_update` / `selective_state_update` set to None so that the reference's own
pure-PyTorch fallback branches run).  Also the reference's `causal_conv1d_update_ref` and `selective_state_update_ref`
on seeded inputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, load_reference, named_fill  # noqa: E402


def main():
    cci, ssi, ms = load_reference()
    ms.causal_conv1d_fn = cci.causal_conv1d_ref
    ms.selective_scan_fn = ssi.selective_scan_ref
    ms.causal_conv1d_update = None
    ms.selective_state_update = None
    torch.manual_seed(0)
    m = ms.Mamba(d_model=12, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=4, layer_idx=0)
    m.load_state_dict(named_fill(m.state_dict()))
    g = torch.Generator().manual_seed(5)
    h = torch.randn(2, 11, 12, generator=g)
    L0 = 8
    params = types.SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0)
    with torch.no_grad():
        out_prefill = m(h[:, :L0], inference_params=params)
        conv0, ssm0 = (t.clone() for t in params.key_value_memory_dict[0])
        outs = []
        for t in range(L0, h.shape[1]):
            params.seqlen_offset = t
            outs.append(m(h[:, t:t + 1], inference_params=params))
        conv1, ssm1 = (t.clone() for t in params.key_value_memory_dict[0])
    arrs = {"h": h, "L0": np.int64(L0), "out_prefill": out_prefill, "conv_state_prefill": conv0, "ssm_state_prefill": ssm0,
            "out_steps": torch.cat(outs, 1), "conv_state_final": conv1, "ssm_state_final": ssm1}
    for k, v in m.state_dict().items():
        arrs["param." + k] = v

    # the two update ops on their own (reference *_ref functions)
    sys.path.insert(0, os.path.join(REF, "mamba"))
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update_ref
    B, D, W, N = 3, 10, 4, 16
    x = torch.randn(B, D, generator=g)
    cs = torch.randn(B, D, W, generator=g)
    wt, bs = torch.randn(D, W, generator=g), torch.randn(D, generator=g)
    cs_in = cs.clone()
    arrs.update({"cu.x": x, "cu.state_in": cs_in, "cu.weight": wt, "cu.bias": bs,
                 "cu.out": cci.causal_conv1d_update_ref(x, cs, wt, bs, "silu"), "cu.state_out": cs})
    st = torch.randn(B, D, N, generator=g)
    dt, z = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    A = -torch.rand(D, N, generator=g) * 2
    Bm, Cm = torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)
    Dv, db = torch.randn(D, generator=g), torch.rand(D, generator=g) - 4.0
    st_in = st.clone()
    arrs.update({"su.state_in": st_in, "su.x": x, "su.dt": dt, "su.z": z, "su.A": A, "su.B": Bm, "su.C": Cm, "su.D": Dv,
                 "su.dt_bias": db,
                 "su.out": selective_state_update_ref(st, x, dt, A, Bm, Cm, Dv, z=z, dt_bias=db, dt_softplus=True),
                 "su.state_out": st})
    np.savez_compressed(os.path.join(HERE, "mamba_decode.npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrs.items()})
    print("wrote mamba_decode.npz:", {k: tuple(np.shape(v)) for k, v in arrs.items() if not k.startswith("param.")})


if __name__ == "__main__":
    main()
