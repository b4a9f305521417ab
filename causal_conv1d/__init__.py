"""Drop-in `causal_conv1d` package (reference: causal-conv1d/causal_conv1d/__init__.py)."""
__version__ = "1.0.0+mi355x"

from causal_conv1d.causal_conv1d_interface import causal_conv1d_fn, causal_conv1d_update  # noqa: F401
