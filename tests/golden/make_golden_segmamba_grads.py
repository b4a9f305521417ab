"""Golden GRADIENTS of the whole network from the REFERENCE's own code (build container only, needs /root/reference, CPU):

    python tests/golden/make_golden_segmamba_grads.py

The reference SegMamba (model_segmamba/segmamba.py on the vendored MONAI blocks and the reference Mamba, CUDA entry points
rebound to the reference's own *_ref functions as in make_golden.py) runs one training-loss forward + backward:
loss = CrossEntropyLoss(model(x), labels) (3_train.py:57-66) on a 32^3 volume, weights from `named_fill`.  Stored in
segmamba_tiny_grads.npz: the loss, the gradient w.r.t. the input (every second voxel), and the gradient of EVERY parameter (fp32; tensors above 8192
elements as a strided sample).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, named_fill  # noqa: E402

if __name__ == "__main__":
    load_reference()
    from model_segmamba.segmamba import SegMamba
    torch.manual_seed(0)
    m = SegMamba(in_chans=4, out_chans=4, depths=[1, 1, 1, 1], feat_size=[48, 8, 16, 32], hidden_size=32)
    m.load_state_dict(named_fill(m.state_dict()))
    m.train()
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(1, 4, 32, 32, 32, generator=g).requires_grad_()
    labels = torch.randint(0, 4, (1, 32, 32, 32), generator=g)
    loss = torch.nn.CrossEntropyLoss()(m(x), labels)
    loss.backward()
    # kept small: dx on every second voxel, gradients of tensors above 8192 elements as every step-th element (flattened)
    out = {"seed": np.array(4321), "loss": loss.detach().numpy(), "dx": x.grad[:, :, ::2, ::2, ::2].numpy().copy()}
    for k, p in m.named_parameters():
        a = p.grad.numpy()
        if a.size > 8192:
            step = -(-a.size // 8192)
            out["grad__" + k] = a.reshape(-1)[::step].copy()
            out["step__" + k] = np.array(step)
        else:
            out["grad__" + k] = a
    np.savez_compressed(os.path.join(HERE, "segmamba_tiny_grads.npz"), **out)
    print("loss", float(loss), "params", sum(1 for _ in m.parameters()),
          "bytes", os.path.getsize(os.path.join(HERE, "segmamba_tiny_grads.npz")))
