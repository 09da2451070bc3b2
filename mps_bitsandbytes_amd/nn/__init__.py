"""Quantized layers (reference: mps_bitsandbytes/nn/__init__.py)."""
from .linear4bit import Linear4bit, Params4bit
from .linear8bit import Linear8bit
from .embedding import Embedding4bit, Embedding8bit, EmbeddingNF4, EmbeddingFP4
from .outlier_aware import OutlierAwareLinear

__all__ = ['Linear4bit', 'Linear8bit', 'Params4bit', 'Embedding4bit', 'Embedding8bit', 'EmbeddingNF4', 'EmbeddingFP4',
           'OutlierAwareLinear']
