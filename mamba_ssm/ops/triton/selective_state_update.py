"""Drop-in path of the reference's `mamba_ssm.ops.triton.selective_state_update` (a Triton kernel there; the MI355X
library's kernel here)."""
from segmamba_amd.selective_state_update import selective_state_update  # noqa: F401
