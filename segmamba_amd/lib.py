"""ctypes binding of libsegmamba_hip.so (C ABI declared in include/segmamba_hip.h).

The product path loads exactly one thing: the HIP library built in-tree by
`__graft_entry__.build()` / `segmamba_amd.build`.  There is no CPU fallback: if
the library is missing, `get_lib()` raises (imports must succeed
*and* the native path must be the one that runs).

PyTorch is used here only as the owner of device memory and streams: tensors are
passed as raw pointers + element strides, the current HIP stream as a handle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libsegmamba_hip.so"
LIB_PATH = os.environ.get("SEGM_LIB_OUT") or os.path.join(_HERE, LIB_NAME)     # SEGM_LIB_OUT: a variant built for an A/B run

SEGM_F32, SEGM_F16, SEGM_BF16 = 0, 1, 2
TIME_FORWARD, TIME_REVERSED, TIME_INTERLEAVED = 0, 1, 2
_DTYPES = {torch.float32: SEGM_F32, torch.float16: SEGM_F16, torch.bfloat16: SEGM_BF16}

_STATUS = {
    -1: "a required pointer is NULL", -2: "bad shape / stride", -3: "dstate out of range ([1, 16] for the scan, [1, 256] for the decode step)",
    -4: "unsupported dtype", -5: "conv width must be in [2, 4]", -6: "workspace missing or too small",
    -7: "unknown time order",
}


class SegmSeq(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("stride_b", C.c_int64), ("stride_t", C.c_int64), ("stride_d", C.c_int64)]


class SegmBC(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("stride_b", C.c_int64), ("stride_g", C.c_int64),
                ("stride_t", C.c_int64), ("stride_n", C.c_int64)]


class ScanFwdArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("dim", C.c_int32), ("dstate", C.c_int32), ("n_groups", C.c_int32),
        ("seqlen", C.c_int64),
        ("dtype", C.c_int32), ("delta_softplus", C.c_int32), ("time_order", C.c_int32), ("nslices", C.c_int32),
        ("chunk", C.c_int32), ("reserved", C.c_int32),
        ("u", SegmSeq), ("delta", SegmSeq), ("z", SegmSeq), ("out", SegmSeq), ("out_z", SegmSeq),
        ("B", SegmBC), ("C", SegmBC),
        ("A", C.c_void_p), ("D", C.c_void_p), ("delta_bias", C.c_void_p),
        ("last_state", C.c_void_p), ("ckpt", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("stream", C.c_void_p),
        ("conv_weight", C.c_void_p), ("conv_bias", C.c_void_p), ("conv_width", C.c_int32), ("reserved2", C.c_int32),
        ("dt_x", C.c_void_p), ("dt_stride_b", C.c_int64), ("dt_stride_t", C.c_int64),
        ("dt_weight", C.c_void_p), ("dt_rank", C.c_int32), ("reserved3", C.c_int32),
    ]


class ScanBwdArgs(C.Structure):
    _fields_ = [
        ("f", ScanFwdArgs),
        ("dout", SegmSeq), ("du", SegmSeq), ("ddelta", SegmSeq), ("dz", SegmSeq),
        ("dB", SegmBC), ("dC", SegmBC),
        ("dA", C.c_void_p), ("dD", C.c_void_p), ("ddelta_bias", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dbc_native", C.c_int32), ("reserved_b", C.c_int32),
    ]


class Conv1dArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("dim", C.c_int32), ("width", C.c_int32), ("silu", C.c_int32),
        ("seqlen", C.c_int64),
        ("dtype", C.c_int32), ("time_order", C.c_int32), ("nslices", C.c_int32), ("reserved", C.c_int32),
        ("x", SegmSeq), ("out", SegmSeq),
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("dout", SegmSeq), ("dx", SegmSeq),
        ("dweight", C.c_void_p), ("dbias", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("stream", C.c_void_p),
    ]


class Conv3dWgradArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32),
        ("width", C.c_int32), ("dtype", C.c_int32), ("dw_dtype", C.c_int32),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_c", C.c_int64), ("x_stride_z", C.c_int64),
        ("x_stride_y", C.c_int64),
        ("dy", C.c_void_p), ("dy_stride_b", C.c_int64), ("dy_stride_c", C.c_int64), ("dy_stride_z", C.c_int64),
        ("dy_stride_y", C.c_int64),
        ("dw", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


class Conv3dFwdArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32),
        ("width", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_int32),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_c", C.c_int64), ("x_stride_z", C.c_int64),
        ("x_stride_y", C.c_int64),
        ("y", C.c_void_p), ("y_stride_b", C.c_int64), ("y_stride_c", C.c_int64), ("y_stride_z", C.c_int64),
        ("y_stride_y", C.c_int64),
        ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("stream", C.c_void_p),
        ("stats_partials", C.c_void_p), ("stats_nparts", C.c_int32), ("reserved", C.c_int32),
    ]


class Conv3dClArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("channels", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("dtype", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_z", C.c_int64), ("x_stride_y", C.c_int64), ("x_stride_x", C.c_int64),
        ("y", C.c_void_p), ("y_stride_b", C.c_int64), ("y_stride_z", C.c_int64), ("y_stride_y", C.c_int64), ("y_stride_x", C.c_int64),
        ("w_image", C.c_void_p), ("bias", C.c_void_p), ("stream", C.c_void_p),
    ]


class Conv3dCubeArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("dtype", C.c_int32), ("flags", C.c_int32), ("nt", C.c_int32), ("splits", C.c_int32),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_c", C.c_int64), ("x_stride_z", C.c_int64), ("x_stride_y", C.c_int64),
        ("y", C.c_void_p), ("y_stride_b", C.c_int64), ("y_stride_c", C.c_int64), ("y_stride_z", C.c_int64), ("y_stride_y", C.c_int64),
        ("w_image", C.c_void_p), ("bias", C.c_void_p), ("workspace", C.c_void_p), ("workspace_elems", C.c_int64), ("stream", C.c_void_p),
        ("stats_partials", C.c_void_p), ("stats_nparts", C.c_int32), ("reserved", C.c_int32),
    ]


class Add3Args(C.Structure):
    _fields_ = [("count", C.c_int64), ("dtype", C.c_int32), ("reserved", C.c_int32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("out", C.c_void_p), ("stream", C.c_void_p)]


class Gather16Args(C.Structure):
    _fields_ = [("count", C.c_int64), ("mode", C.c_int32), ("reserved", C.c_int32),
                ("src", C.c_void_p), ("map", C.c_void_p), ("out", C.c_void_p), ("stream", C.c_void_p)]


class InstNormFwdArgs(C.Structure):
    _fields_ = [
        ("instances", C.c_int32), ("dtype", C.c_int32), ("act", C.c_int32), ("reserved", C.c_int32),
        ("spatial", C.c_int64), ("slope", C.c_float), ("eps", C.c_float),
        ("x", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("x_instance_stride", C.c_int64), ("residual_instance_stride", C.c_int64), ("y_instance_stride", C.c_int64),
        ("stats_partials", C.c_void_p), ("stats_nparts", C.c_int32), ("reserved2", C.c_int32),
    ]


class InstNormBwdArgs(C.Structure):
    _fields_ = [
        ("instances", C.c_int32), ("dtype", C.c_int32), ("act", C.c_int32), ("reserved", C.c_int32),
        ("spatial", C.c_int64), ("slope", C.c_float), ("reserved2", C.c_float),
        ("x", C.c_void_p), ("dy", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("dx", C.c_void_p), ("dresidual", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("x_instance_stride", C.c_int64), ("dy_instance_stride", C.c_int64), ("y_instance_stride", C.c_int64),
        ("dx_instance_stride", C.c_int64), ("dresidual_instance_stride", C.c_int64),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("channels", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32),
        ("spatial", C.c_int64), ("eps", C.c_float), ("reserved2", C.c_float),
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p),
        ("rstd", C.c_void_p), ("dy", C.c_void_p), ("dx", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


class SgdArgs(C.Structure):
    _fields_ = [
        ("ntensors", C.c_int32), ("nesterov", C.c_int32),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("momenta", C.c_void_p), ("numel", C.c_void_p),
        ("lr", C.c_float), ("momentum", C.c_float), ("weight_decay", C.c_float), ("max_norm", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("loss_scale", C.c_void_p), ("found_inf", C.c_void_p),
    ]


class CrossEntropyArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("classes", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32),
        ("spatial", C.c_int64), ("ignore_index", C.c_int64),
        ("logits", C.c_void_p), ("labels", C.c_void_p), ("dlogits", C.c_void_p), ("loss_partial", C.c_void_p),
        ("count_partial", C.c_void_p), ("stream", C.c_void_p),
    ]


class Conv1dUpdateArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("dim", C.c_int32), ("width", C.c_int32), ("silu", C.c_int32),
        ("dtype", C.c_int32), ("reserved", C.c_int32),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_d", C.c_int64),
        ("conv_state", C.c_void_p), ("state_stride_b", C.c_int64), ("state_stride_d", C.c_int64), ("state_stride_w", C.c_int64),
        ("out", C.c_void_p), ("out_stride_b", C.c_int64), ("out_stride_d", C.c_int64),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("stream", C.c_void_p),
    ]


class StateUpdateArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("dim", C.c_int32), ("dstate", C.c_int32), ("dt_softplus", C.c_int32),
        ("dtype", C.c_int32), ("state_dtype", C.c_int32),
        ("state", C.c_void_p), ("state_stride_b", C.c_int64), ("state_stride_d", C.c_int64), ("state_stride_n", C.c_int64),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_d", C.c_int64),
        ("dt", C.c_void_p), ("dt_stride_b", C.c_int64), ("dt_stride_d", C.c_int64),
        ("z", C.c_void_p), ("z_stride_b", C.c_int64), ("z_stride_d", C.c_int64),
        ("out", C.c_void_p), ("out_stride_b", C.c_int64), ("out_stride_d", C.c_int64),
        ("B", C.c_void_p), ("B_stride_b", C.c_int64), ("B_stride_n", C.c_int64),
        ("C", C.c_void_p), ("C_stride_b", C.c_int64), ("C_stride_n", C.c_int64),
        ("A", C.c_void_p), ("D", C.c_void_p), ("dt_bias", C.c_void_p), ("stream", C.c_void_p),
    ]


class LinearArgs(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("k", C.c_int32), ("n", C.c_int32), ("dtype", C.c_int32), ("accumulate", C.c_int32),
        ("x", C.c_void_p), ("x_stride_row", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("y", C.c_void_p), ("y_stride_row", C.c_int64), ("stream", C.c_void_p),
    ]


class PointwiseArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("dtype", C.c_int32), ("accumulate", C.c_int32),
        ("spatial", C.c_int64),
        ("x", C.c_void_p), ("x_stride_b", C.c_int64), ("x_stride_c", C.c_int64),
        ("w", C.c_void_p), ("w_stride", C.c_int32), ("bias", C.c_void_p),
        ("y", C.c_void_p), ("y_stride_b", C.c_int64), ("y_stride_c", C.c_int64), ("stream", C.c_void_p),
    ]


class StemArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cout", C.c_int32), ("din", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("dtype", C.c_int32),
        ("x4", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("stream", C.c_void_p),
        ("kernel_size", C.c_int32), ("stride", C.c_int32), ("y_channel_stride", C.c_int64),
    ]


class StemWgradArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("cout", C.c_int32), ("din", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("dtype", C.c_int32),
        ("x4", C.c_void_p), ("dy", C.c_void_p), ("dw_packed", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("kernel_size", C.c_int32), ("stride", C.c_int32), ("dy_channel_stride", C.c_int64),
    ]


class WgradGemmArgs(C.Structure):
    _fields_ = [
        ("layout", C.c_int32), ("dtype", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int64),
        ("batch", C.c_int32), ("reserved", C.c_int32),
        ("a", C.c_void_p), ("a_stride_row", C.c_int64), ("a_stride_batch", C.c_int64),
        ("b", C.c_void_p), ("b_stride_row", C.c_int64), ("b_stride_batch", C.c_int64),
        ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


class D2sArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("channels", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("dtype", C.c_int32), ("direction", C.c_int32), ("reserved", C.c_int32),
        ("blk", C.c_void_p), ("vol", C.c_void_p),
        ("vol_stride_b", C.c_int64), ("vol_stride_c", C.c_int64), ("vol_stride_z", C.c_int64), ("vol_stride_y", C.c_int64),
        ("stream", C.c_void_p),
    ]


class TransposeArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("dtype", C.c_int32),
        ("in_", C.c_void_p), ("add", C.c_void_p), ("out", C.c_void_p), ("stream", C.c_void_p),
    ]


EXPORTS = (
    "segm_selective_scan_fwd", "segm_selective_scan_fwd_workspace_bytes", "segm_selective_scan_ckpt_bytes",
    "segm_selective_scan_default_chunk", "segm_selective_scan_bwd", "segm_selective_scan_bwd_workspace_bytes",
    "segm_selective_scan_bwd_deterministic",
    "segm_selective_scan_fwd_multi", "segm_selective_scan_bwd_multi", "segm_causal_conv1d_fwd_multi", "segm_causal_conv1d_bwd_multi",
    "segm_causal_conv1d_fwd", "segm_causal_conv1d_bwd", "segm_causal_conv1d_bwd_workspace_bytes",
    "segm_conv3d_k3_wgrad", "segm_conv3d_k3_wgrad_workspace_bytes", "segm_conv3d_k3_fwd", "segm_conv3d_k3_fwd_stats_parts",
    "segm_conv3d_k3_fwd_cl", "segm_conv3d_k3_cl_pack_index", "segm_add3",
    "segm_conv3d_k3_cube_fwd", "segm_conv3d_k3_cube_plan", "segm_conv3d_k3_cube_pack_index",
    "segm_conv3d_k3_cube_wgrad", "segm_conv3d_k3_cube_wgrad_workspace_bytes", "segm_gather16",
    "segm_conv3d_k3_cube_pack_multi", "segm_conv3d_k3_cube_stats_parts",
    "segm_instnorm_fwd", "segm_instnorm_bwd", "segm_instnorm_workspace_bytes", "segm_transpose_add", "segm_depth_to_space2",
    "segm_layernorm_tokens_fwd", "segm_layernorm_tokens_bwd", "segm_layernorm_tokens_workspace_bytes",
    "segm_sgd_clip_step", "segm_sgd_clip_step_workspace_bytes", "segm_cross_entropy", "segm_cross_entropy_partials",
    "segm_causal_conv1d_update", "segm_selective_state_update", "segm_linear_rows", "segm_pointwise_cf", "segm_stem_conv_fwd",
    "segm_stem_conv_wgrad", "segm_stem_conv_wgrad_workspace_bytes", "segm_stem_conv_wgrad_workspace_bytes2", "segm_wgrad_gemm", "segm_wgrad_gemm_workspace_bytes",
    "segm_skinny_tn", "segm_skinny_tn_workspace_bytes", "segm_channel_sum", "segm_channel_sum_workspace_bytes", "segm_selective_scan_regular_shape",
    "segm_abi_version", "segm_status_string",
)


class SkinnyTnArgs(C.Structure):
    _fields_ = [("k", C.c_int64), ("m", C.c_int32), ("n", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32),
                ("wide", C.c_void_p), ("wide_stride_row", C.c_int64), ("skinny", C.c_void_p), ("skinny_stride_row", C.c_int64),
                ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p)]


class ChannelSumArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("stride_batch", C.c_int64), ("stride_channel", C.c_int64), ("spatial", C.c_int64),
                ("batch", C.c_int32), ("channels", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32),
                ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p)]


def header_abi_version() -> int:
    """SEGM_ABI_VERSION as include/segmamba_hip.h declares it (what a freshly built library must report)"""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "segmamba_hip.h")
    m = re.search(r"#define\s+SEGM_ABI_VERSION\s+(\d+)", open(hdr).read())
    if not m:
        raise RuntimeError("include/segmamba_hip.h does not define SEGM_ABI_VERSION")
    return int(m.group(1))


class SegmLib:
    """A loaded C-ABI library (the HIP one in production; tests may load the CPU emulation build)."""

    def __init__(self, path: str):
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        self.missing = [n for n in EXPORTS if not hasattr(d, n)]

        def sig(name, argtypes, restype):
            if hasattr(d, name):
                fn = getattr(d, name)
                fn.argtypes, fn.restype = argtypes, restype

        sig("segm_selective_scan_fwd", [C.POINTER(ScanFwdArgs)], C.c_int)
        sig("segm_selective_scan_bwd", [C.POINTER(ScanBwdArgs)], C.c_int)
        sig("segm_selective_scan_bwd_deterministic", [C.POINTER(ScanBwdArgs)], C.c_int)
        sig("segm_selective_scan_fwd_multi", [C.POINTER(ScanFwdArgs), C.c_int32], C.c_int)
        sig("segm_selective_scan_bwd_multi", [C.POINTER(ScanBwdArgs), C.c_int32], C.c_int)
        for n in ("segm_selective_scan_fwd_workspace_bytes", "segm_selective_scan_bwd_workspace_bytes"):
            sig(n, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32], C.c_size_t)
        sig("segm_selective_scan_ckpt_bytes", [C.c_int32, C.c_int32, C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_selective_scan_default_chunk", [C.c_int32, C.c_int32, C.c_int64], C.c_int32)
        sig("segm_causal_conv1d_fwd", [C.POINTER(Conv1dArgs)], C.c_int)
        sig("segm_causal_conv1d_bwd", [C.POINTER(Conv1dArgs)], C.c_int)
        sig("segm_causal_conv1d_fwd_multi", [C.POINTER(Conv1dArgs), C.c_int32], C.c_int)
        sig("segm_causal_conv1d_bwd_multi", [C.POINTER(Conv1dArgs), C.c_int32], C.c_int)
        sig("segm_causal_conv1d_bwd_workspace_bytes", [C.c_int32, C.c_int32, C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_conv3d_k3_wgrad", [C.POINTER(Conv3dWgradArgs)], C.c_int)
        sig("segm_conv3d_k3_wgrad_workspace_bytes", [C.c_int32] * 6, C.c_size_t)
        sig("segm_conv3d_k3_fwd", [C.POINTER(Conv3dFwdArgs)], C.c_int)
        sig("segm_conv3d_k3_fwd_stats_parts", [C.c_int32] * 6, C.c_int32)
        sig("segm_conv3d_k3_fwd_cl", [C.POINTER(Conv3dClArgs)], C.c_int)
        sig("segm_conv3d_k3_cl_pack_index", [C.c_void_p, C.c_int64], C.c_int)
        sig("segm_conv3d_k3_cube_fwd", [C.POINTER(Conv3dCubeArgs)], C.c_int)
        sig("segm_conv3d_k3_cube_plan", [C.c_int32] * 6 + [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)], C.c_int)
        sig("segm_conv3d_k3_cube_pack_index", [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32], C.c_int)
        sig("segm_conv3d_k3_cube_wgrad", [C.POINTER(Conv3dWgradArgs)], C.c_int)
        sig("segm_conv3d_k3_cube_stats_parts", [C.c_int32] * 4, C.c_int32)
        sig("segm_conv3d_k3_cube_pack_multi", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p], C.c_int)
        sig("segm_conv3d_k3_cube_wgrad_workspace_bytes", [C.c_int32] * 6, C.c_size_t)
        sig("segm_add3", [C.POINTER(Add3Args)], C.c_int)
        sig("segm_gather16", [C.POINTER(Gather16Args)], C.c_int)
        sig("segm_instnorm_fwd", [C.POINTER(InstNormFwdArgs)], C.c_int)
        sig("segm_instnorm_bwd", [C.POINTER(InstNormBwdArgs)], C.c_int)
        sig("segm_instnorm_workspace_bytes", [C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_transpose_add", [C.POINTER(TransposeArgs)], C.c_int)
        sig("segm_depth_to_space2", [C.POINTER(D2sArgs)], C.c_int)
        sig("segm_layernorm_tokens_fwd", [C.POINTER(LayerNormArgs)], C.c_int)
        sig("segm_layernorm_tokens_bwd", [C.POINTER(LayerNormArgs)], C.c_int)
        sig("segm_layernorm_tokens_workspace_bytes", [C.c_int32, C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_sgd_clip_step", [C.POINTER(SgdArgs)], C.c_int)
        sig("segm_sgd_clip_step_workspace_bytes", [C.c_int32, C.c_void_p], C.c_size_t)
        sig("segm_cross_entropy", [C.POINTER(CrossEntropyArgs)], C.c_int)
        sig("segm_cross_entropy_partials", [C.c_int32, C.c_int64], C.c_int32)
        sig("segm_causal_conv1d_update", [C.POINTER(Conv1dUpdateArgs)], C.c_int)
        sig("segm_selective_state_update", [C.POINTER(StateUpdateArgs)], C.c_int)
        sig("segm_linear_rows", [C.POINTER(LinearArgs)], C.c_int)
        sig("segm_pointwise_cf", [C.POINTER(PointwiseArgs)], C.c_int)
        sig("segm_stem_conv_fwd", [C.POINTER(StemArgs)], C.c_int)
        sig("segm_stem_conv_wgrad", [C.POINTER(StemWgradArgs)], C.c_int)
        sig("segm_stem_conv_wgrad_workspace_bytes", [C.c_int32] * 4, C.c_size_t)
        sig("segm_stem_conv_wgrad_workspace_bytes2", [C.c_int32] * 6, C.c_size_t)
        sig("segm_wgrad_gemm", [C.POINTER(WgradGemmArgs)], C.c_int)
        sig("segm_wgrad_gemm_workspace_bytes", [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32], C.c_size_t)
        sig("segm_skinny_tn", [C.POINTER(SkinnyTnArgs)], C.c_int)
        sig("segm_skinny_tn_workspace_bytes", [C.c_int32, C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_selective_scan_regular_shape", [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32], C.c_int32)
        sig("segm_channel_sum", [C.POINTER(ChannelSumArgs)], C.c_int)
        sig("segm_channel_sum_workspace_bytes", [C.c_int32, C.c_int32, C.c_int64], C.c_size_t)
        sig("segm_abi_version", [], C.c_int)
        sig("segm_status_string", [C.c_int], C.c_char_p)

    def check(self, rc: int, what: str) -> None:
        if rc == 0:
            return
        if rc < 0:
            msg = _STATUS.get(rc, "unknown status")
            raise RuntimeError(f"{what}: {msg} (status {rc})")
        raise RuntimeError(f"{what}: HIP error {rc}")


_lib: Optional[SegmLib] = None


def get_lib() -> SegmLib:
    """The HIP library.  Raises if it has not been built - the product never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  segmamba_amd has no CPU / PyTorch fallback for its kernels.")
        _lib = SegmLib(LIB_PATH)
    return _lib


# ---------------------------------------------------------------------------------------------------------
# tensor -> view descriptors
# ---------------------------------------------------------------------------------------------------------
def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}: expected float32, float16 or bfloat16") from None


def seq_view(t: Optional[torch.Tensor], channel_last: bool) -> SegmSeq:
    """Descriptor of a 3-D tensor: (B, L, D) if channel_last else (B, D, L)."""
    if t is None:
        return SegmSeq(None, 0, 0, 0)
    assert t.dim() == 3
    sb, s1, s2 = t.stride()
    return SegmSeq(t.data_ptr(), sb, s1, s2) if channel_last else SegmSeq(t.data_ptr(), sb, s2, s1)


def bc_view(t: torch.Tensor, channel_last: bool) -> SegmBC:
    """Descriptor of B / C: (B, L, G, N) if channel_last else (B, G, N, L)."""
    assert t.dim() == 4
    s = t.stride()
    if channel_last:
        return SegmBC(t.data_ptr(), s[0], s[2], s[1], s[3])
    return SegmBC(t.data_ptr(), s[0], s[1], s[3], s[2])


def on_device(t: torch.Tensor) -> bool:
    """Whether `t` takes the library's kernels (CUDA tensors do).  Host modules ask this instead of `t.is_cuda` so that the
    CPU tests can send CPU tensors through the emulated kernels; the product never patches it."""
    return t.is_cuda


def stream_handle(t: torch.Tensor) -> Optional[int]:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def fptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()
