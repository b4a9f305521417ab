"""-m gpu: the drop-in sentence of BASELINE.json ("drops into 3_train.py / 0_inference.py unchanged"), executed.

`0_inference.py` of the reference (lines 1-16) imports `from model_segmamba.segmamba import SegMamba`, builds
SegMamba(in_chans=4, out_chans=4, depths=[2,2,2,2], feat_size=[48, 96, 192, 384]).cuda() and calls it - fp32, gradients
enabled, no autocast - on torch.rand(1, 4, 128, 128, 128).cuda(), then prints out.shape.  The reference tree is not on the GPU
box, so the same statements run here in a fresh interpreter whose sys.path has only this repository ahead of site-packages
(the reference's script finds these packages first).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = '''
import torch
from model_segmamba.segmamba import SegMamba
import model_segmamba.segmamba, mamba_ssm, causal_conv1d
for mod in (model_segmamba.segmamba, mamba_ssm, causal_conv1d):
    assert mod.__file__.startswith(ROOT), mod.__file__          # the drop-in packages, not some other install
t1 = torch.rand(1, 4, 128, 128, 128).cuda()
model = SegMamba(in_chans=4,
                 out_chans=4,
                 depths=[2,2,2,2],
                 feat_size=[48, 96, 192, 384]).cuda()
out = model(t1)
print(out.shape)
assert out.dtype == torch.float32 and out.requires_grad and torch.isfinite(out).all()
# fp32 at C = 384 leaves the fused LayerNorm kernel (fp32 <= 192 channels): the full-size network must still agree with
# itself under bf16 autocast to bf16 accuracy
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    out16 = model(t1)
err = (out16.float() - out).abs().max().item() / max(1.0, out.abs().max().item())
print("bf16 vs fp32 rel err", err)
assert err < 5e-2
out.float().mean().backward()                                   # gradients enabled in the script: the graph must be usable
assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
import segmamba_amd.lib as L
print("native library:", L.get_lib().path)
'''


def test_0_inference_body_runs_through_the_dropin_packages():
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + BODY], capture_output=True, text=True, timeout=900,
                       env=env, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "torch.Size([1, 4, 128, 128, 128])" in r.stdout
    assert "libsegmamba_hip.so" in r.stdout


def test_kernels_follow_the_tensors_device_not_the_current_device():
    """a model on cuda:1 while the process's current device is cuda:0 (needs two GPUs; skipped on a 1-GPU box)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from oracle import ref_ops
    from segmamba_amd import lib as L, ops_raw
    from tests import helpers as H
    torch.cuda.set_device(0)
    c = H.scan_case(2, 32, 16, 256, seed=9)
    ref = H.scan_oracle(c, want_grads=False)
    res = H.run_scan(L.get_lib(), c, "cuda:1", True, backward=False)
    assert res["out"].device.index == 1 and torch.cuda.current_device() == 0
    H.assert_close(res["out"], ref["out"], 1e-3, 1e-3, "scan on cuda:1 with cuda:0 current")


def test_whole_mamba_block_follows_the_tensors_device():
    """the three-direction entry points (`*_multi`) and `channel_sum` take the device of their tensors too - a whole
    Mamba(v3) block, forward and backward, on cuda:1 while cuda:0 is current equals the same block on cuda:0 (two GPUs needed)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from mamba_ssm import Mamba
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    m0 = Mamba(d_model=48, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=8).to("cuda:0")
    m1 = Mamba(d_model=48, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=8).to("cuda:1")
    m1.load_state_dict(m0.state_dict())
    x = torch.randn(2, 512, 48)
    outs = []
    for m, dev in ((m0, "cuda:0"), (m1, "cuda:1")):
        xi = x.to(dev).requires_grad_()
        y = m(xi)
        y.square().mean().backward()
        assert torch.cuda.current_device() == 0 and y.device == torch.device(dev)
        outs.append((y.detach().cpu(), xi.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()}))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)
    for k in outs[0][2]:
        assert torch.allclose(outs[0][2][k], outs[1][2][k], rtol=1e-4, atol=1e-6), k
