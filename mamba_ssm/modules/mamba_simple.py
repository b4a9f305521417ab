"""Drop-in for reference mamba/mamba_ssm/modules/mamba_simple.py (class `Mamba`)."""
from segmamba_amd.mamba_simple import Mamba  # noqa: F401
