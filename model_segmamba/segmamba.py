"""Drop-in for reference model_segmamba/segmamba.py: `from model_segmamba.segmamba import SegMamba`
(0_inference.py:4, 3_train.py:39) resolves to the MI355X implementation."""
from segmamba_amd.segmamba import GSC, MambaEncoder, MambaLayer, MlpChannel, SegMamba  # noqa: F401
