"""Golden GRADIENTS of the whole network from the REFERENCE's own code (build container only, needs /root/reference, CPU):

    python tests/golden/make_golden_segmamba_grads.py

The reference SegMamba (model_segmamba/segmamba.py on the vendored MONAI blocks and the reference Mamba, CUDA entry points
rebound to the reference's own *_ref functions as in make_golden.py) runs one training-loss forward + backward:
loss = CrossEntropyLoss(model(x), labels) (3_train.py:57-66) on a 32^3 volume, weights from `named_fill`, once in fp32 and
once with the modules in fp64.  Stored in segmamba_tiny_grads.npz: the loss, the gradient w.r.t. the input (every second
voxel) and the gradient of EVERY parameter from the fp64 run (tensors above 8192 elements as a strided sample), plus, per
tensor, the distance of the reference's own fp32 gradients from them ("noise__*").
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, named_fill  # noqa: E402

def run_reference(dtype):
    from model_segmamba.segmamba import SegMamba
    torch.manual_seed(0)
    m = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 8, 16, 32], hidden_size=32)
    m.load_state_dict(named_fill(m.state_dict()))
    m = m.to(dtype).train()
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(1, 4, 32, 32, 32, generator=g).to(dtype).requires_grad_()
    labels = torch.randint(0, 4, (1, 32, 32, 32), generator=g)
    loss = torch.nn.CrossEntropyLoss()(m(x), labels)
    loss.backward()
    return float(loss), x.grad.double().numpy(), {k: p.grad.double().numpy() for k, p in m.named_parameters()}


if __name__ == "__main__":
    load_reference()
    # The reference in fp32 (what 3_train.py runs) and again with every module in fp64 (its selective_scan_ref still
    # computes in fp32 internally, selective_scan_interface.py:97-99).  The tiny network is ill-conditioned - InstanceNorm over
    # 2^3 .. 4^3 voxels - so the reference's OWN fp32 gradients sit up to ~1 % of max|grad| away from its fp64 ones; that
    # measured distance ("noise__*", max abs per tensor) is stored and is what the comparison allows on top of 1e-3.
    loss32, dx32, g32 = run_reference(torch.float32)
    loss64, dx64, g64 = run_reference(torch.float64)
    sub = (slice(None), slice(None), slice(None, None, 2), slice(None, None, 2), slice(None, None, 2))
    out = {"seed": np.array(4321), "loss": np.array(loss64), "loss_fp32": np.array(loss32),
           "dx": dx64[sub].astype(np.float32), "noise__dx": np.array(np.abs(dx32 - dx64).max())}
    for k in g64:
        a = g64[k]
        out["noise__" + k] = np.array(np.abs(g32[k] - a).max())
        if a.size > 8192:
            step = -(-a.size // 8192)
            out["grad__" + k] = a.reshape(-1)[::step].astype(np.float32)
            out["step__" + k] = np.array(step)
        else:
            out["grad__" + k] = a.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "segmamba_tiny_grads.npz"), **out)
    print("loss", loss32, loss64, "dx noise", float(out["noise__dx"]), "of", float(np.abs(dx64).max()),
          "bytes", os.path.getsize(os.path.join(HERE, "segmamba_tiny_grads.npz")))
