"""CPU ORACLE (test infrastructure): the whole SegMamba network on the reference's pure-PyTorch CPU path.

`segmamba_amd.SegMamba` on CPU tensors runs its convolution / normalisation layers through ATen (the library kernels take
CUDA tensors only) but has no CPU Mamba block - the product has no fallback.  This module supplies the missing piece FOR
MEASUREMENT AND TESTS ONLY: every `Mamba.forward` is replaced by the oracle's restatement of the reference's v3 forward
(`ref_ops.mamba_v3_forward_ref`: causal_conv1d_ref + selective_scan_ref, mamba_simple.py:204-264), which is what
BASELINE.json's config 0 ("selective_scan_ref pure-PyTorch CPU path") and bench.py's cpu_baseline time.
"""
from __future__ import annotations

import types

import torch

from . import ref_ops


def cpu_reference_segmamba(**kw):
    from segmamba_amd.mamba_simple import Mamba
    from segmamba_amd.segmamba import SegMamba
    net = SegMamba(**kw)

    def fwd(self, hidden_states, inference_params=None):
        p = {k: v for k, v in self.named_parameters()}
        return ref_ops.mamba_v3_forward_ref(hidden_states, p, self.nslices)

    for m in net.modules():
        if isinstance(m, Mamba):
            m.forward = types.MethodType(fwd, m)
    return net
