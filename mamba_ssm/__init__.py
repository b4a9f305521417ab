"""Drop-in `mamba_ssm` package (reference: mamba/mamba_ssm/__init__.py): the names SegMamba imports, backed by
the MI355X kernels in `segmamba_amd`.  The language-model stack of the reference package (`models/`, `utils/`)
is outside the hot path and is not provided."""
__version__ = "1.0.1+mi355x"

from segmamba_amd.selective_scan_interface import (  # noqa: F401
    selective_scan_fn, mamba_inner_fn, bimamba_inner_fn, mamba_inner_fn_no_out_proj,
)
from segmamba_amd.mamba_simple import Mamba  # noqa: F401
