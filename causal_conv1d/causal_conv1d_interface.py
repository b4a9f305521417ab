"""Drop-in for reference causal-conv1d/causal_conv1d/causal_conv1d_interface.py (same public names)."""
from segmamba_amd.causal_conv1d_interface import CausalConv1dFn, causal_conv1d_fn, causal_conv1d_update  # noqa: F401
