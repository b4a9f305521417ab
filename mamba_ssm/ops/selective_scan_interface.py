"""Drop-in for reference mamba/mamba_ssm/ops/selective_scan_interface.py (same public names)."""
from segmamba_amd.selective_scan_interface import (  # noqa: F401
    SelectiveScanFn, selective_scan_fn, MambaInnerCore, mamba_inner_fn, bimamba_inner_fn,
    mamba_inner_fn_no_out_proj,
)
