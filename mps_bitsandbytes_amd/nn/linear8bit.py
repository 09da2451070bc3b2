"""
Linear8bit — row-wise INT8 linear layer on MI355X.

Public surface of the reference module (mps_bitsandbytes/nn/linear8bit.py:15-166): buffers `weight_int8` [N, K] int8
and `weight_scales` [N] f32 (the row absmax), `from_linear`, the dequantized-weight cache API and `device`.
The forward never materialises the dequantized weight: `functional.linear_int8` decodes the int8 rows inside the GEMM
(skinny MFMA, split-K or the 256 x 256 LDS-DMA kernel), with the reference's dequantize_rowwise -> F.linear result.
"""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F
from ._base import QuantizedModule, compute_dtype_for, source_device


class Linear8bit(QuantizedModule):
    _anchor = 'weight_int8'

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 use_cache: bool = True, compute_dtype: torch.dtype = torch.float16):
        super().__init__()
        self._init_linear(in_features, out_features, compute_dtype)
        self.use_cache = use_cache
        self.register_buffer('weight_int8', torch.zeros(out_features, in_features, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(out_features, dtype=torch.float32, device=device))
        self._init_bias(bias, device)
        self._weight_cache: Optional[Tensor] = None

    def forward(self, x: Tensor) -> Tensor:
        return F.linear_int8(x, self.weight_int8, self.weight_scales, self.bias, dtype=self.compute_dtype)

    # dequantized view for LoRA merges / debugging (reference :70-89); forward() does not use it
    def _get_weight(self) -> Tensor:
        cached = self._weight_cache if self.use_cache else None
        if cached is None:
            cached = F.dequantize_rowwise(self.weight_int8, self.weight_scales, dtype=self.compute_dtype)
            if self.use_cache:
                self._weight_cache = cached
        return cached

    def clear_cache(self):
        self._weight_cache = None

    @classmethod
    def from_linear(cls, linear: nn.Linear, device=None, use_cache: bool = True,
                    compute_dtype: Optional[torch.dtype] = None) -> 'Linear8bit':
        """Quantize an nn.Linear on `device` with the HIP kernel (reference :104-151)."""
        device = source_device(linear.weight, device)
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=device,
                    use_cache=use_cache, compute_dtype=compute_dtype_for(linear.weight.dtype, compute_dtype))
        q, absmax = F.quantize_rowwise(linear.weight.data.to(device))
        layer.weight_int8.copy_(q)
        layer.weight_scales.copy_(absmax.float())
        layer._copy_bias_from(linear, device)
        return layer

    def extra_repr(self) -> str:
        return self._repr_core()
