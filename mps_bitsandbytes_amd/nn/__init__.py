"""Quantized linear layers (reference: mps_bitsandbytes/nn/__init__.py)."""
from .linear4bit import Linear4bit, Params4bit
from .linear8bit import Linear8bit

__all__ = ['Linear4bit', 'Linear8bit', 'Params4bit']
