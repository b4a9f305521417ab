"""The two native modules the REFERENCE's own Python imports, bound to libsegmamba_hip.so.

The reference's `mamba_ssm/ops/selective_scan_interface.py:9-11` does `import causal_conv1d_cuda` and
`import selective_scan_cuda` - pybind11 extensions built from mamba/csrc/selective_scan/selective_scan.cpp:494-497 and
causal-conv1d/csrc/causal_conv1d.cpp:329-333.  A maintainer who keeps the reference's Python and swaps only the native
layer calls `install()` (or puts this directory on sys.path) before importing the reference:

    from segmamba_amd import native_stubs; native_stubs.install()
    import mamba_ssm.ops.selective_scan_interface        # the reference's file, unchanged

Same function names, argument lists, return lists and in-place conventions as the C++ entry points.
tests/test_reference_bindings.py runs the reference's own autograd Functions on top of these modules.
"""
import sys


def install():
    from . import causal_conv1d_cuda, selective_scan_cuda
    sys.modules["selective_scan_cuda"] = selective_scan_cuda
    sys.modules["causal_conv1d_cuda"] = causal_conv1d_cuda
    return selective_scan_cuda, causal_conv1d_cuda
