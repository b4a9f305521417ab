"""MI355X-native hot path of SegMamba: host side of libsegmamba_hip.so."""
import os as _os

# MIOpen's "find" otherwise times - and for some 3-D shapes picks - its naive reference convolution solvers (0.4 s per call at
# 128^3, profiles/r01_bench_step_kernels.txt).  The convolutions that stay on MIOpen (7^3 stem, 16^3 / 8^3 layers, fp32
# inference as in the reference's 0_inference.py) should not; set before the first convolution runs, never overriding the user.
for _k in ("FWD", "BWD", "WRW"):
    _os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
