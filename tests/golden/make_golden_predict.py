"""Golden fixtures of the prediction path (sliding window + mirror TTA), generated from the REFERENCE's own code:
MONAI `SlidingWindowInferer` (vendored under /root/reference/monai) driven by the reference `Predictor`
(light_training/prediction.py).  Run in the build container only:  python tests/golden/make_golden_predict.py

The reference module imports SimpleITK / skimage / its resampling package at import time (not installed here, not used by
the functions called): they are stubbed with empty modules, the functions themselves run unchanged.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def toy_net(cin: int, cout: int) -> torch.nn.Module:
    """a fixed 3x3x3 conv + tanh: position dependent enough to expose stitching / mirroring mistakes"""
    net = torch.nn.Conv3d(cin, cout, 3, padding=1)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        net.weight.copy_(0.3 * torch.randn(net.weight.shape, generator=g))
        net.bias.copy_(0.1 * torch.randn(net.bias.shape, generator=g))
    return torch.nn.Sequential(net, torch.nn.Tanh()).eval()


CASES = {
    # name: (input shape, roi, sw_batch, overlap, mode, mirror axes)
    "gauss_half": ((1, 2, 20, 27, 33), (16, 16, 16), 2, 0.5, "gaussian", [0, 1, 2]),
    "const_quarter": ((2, 2, 18, 16, 25), (16, 16, 16), 1, 0.25, "constant", [0, 2]),
    "smaller_than_roi": ((1, 2, 10, 16, 13), (16, 16, 16), 4, 0.5, "gaussian", None),
}

if __name__ == "__main__":
    for name in ("SimpleITK", "skimage", "skimage.measure"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    stub = types.ModuleType("light_training.preprocessing.resampling.default_resampling")
    stub.resample_data_or_seg_to_shape = None
    sys.modules["light_training.preprocessing.resampling.default_resampling"] = stub
    sys.path.insert(0, REF)
    from monai.inferers import SlidingWindowInferer
    from light_training.prediction import Predictor

    out = {}
    for name, (shape, roi, swb, ov, mode, axes) in CASES.items():
        x = torch.rand(shape, generator=torch.Generator().manual_seed(sum(shape)))
        net = toy_net(shape[1], 3)
        inferer = SlidingWindowInferer(roi_size=roi, sw_batch_size=swb, overlap=ov, mode=mode)
        with torch.no_grad():
            out[name + "_window"] = inferer(x, net).numpy()
        out[name + "_tta"] = Predictor(window_infer=inferer, mirror_axes=axes).maybe_mirror_and_predict(x, net).numpy()
    props = {"shape_after_cropping_before_resample": [11, 9, 14], "shape_before_cropping": [20, 18, 22],
             "bbox_used_for_cropping": [[3, 14], [5, 14], [2, 16]]}
    logits = torch.rand(1, 3, 8, 8, 8, generator=torch.Generator().manual_seed(3))
    raw = Predictor.predict_raw_probability(logits, props).float()
    out["raw_probability"] = raw.numpy()
    out["noncrop"] = Predictor.predict_noncrop_probability(raw.argmax(0).to(torch.uint8), props)
    np.savez_compressed(os.path.join(HERE, "predict.npz"), **out)
    print({k: v.shape for k, v in out.items()})
