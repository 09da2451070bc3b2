"""
Quantized layers of the MI355X backend.

Each layer keeps the reference's public surface (mps_bitsandbytes/nn/*.py) and runs its forward through one entry
point of the HIP library: Linear4bit -> matmul_4bit, Linear8bit -> linear_int8, LinearFP8 -> matmul_fp8_e4m3, OutlierAwareLinear -> outlier_linear,
Embedding4bit / Embedding8bit -> embedding_4bit / embedding_8bit.  `_base.py` holds what they share.
"""
from .embedding import Embedding4bit, Embedding8bit, EmbeddingFP4, EmbeddingNF4
from .linear4bit import Linear4bit, Params4bit
from .linear8bit import Linear8bit
from .linear_fp8 import LinearFP8
from .outlier_aware import OutlierAwareLinear

__all__ = sorted(['Linear4bit', 'Params4bit', 'Linear8bit', 'LinearFP8', 'OutlierAwareLinear',
                  'Embedding4bit', 'Embedding8bit', 'EmbeddingNF4', 'EmbeddingFP4'])
