"""
Linear8bit — rowwise-INT8 quantized linear layer on MI355X.

Same constructor, buffers (`weight_int8` [N,K] int8, `weight_scales` [N] f32), `from_linear`,
cache API and `device` property as the reference (mps_bitsandbytes/nn/linear8bit.py:15-166).
`forward` does not materialise the dequantized weight: the int8 rows are decoded inside the
MFMA GEMM's B-tile producer (functional.linear_int8), giving the same result as the reference's
dequantize_rowwise -> F.linear.
"""
from typing import Optional

import torch
from torch import nn, Tensor

from ..functional import quantize_rowwise, dequantize_rowwise, linear_int8


class Linear8bit(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 use_cache: bool = True, compute_dtype: torch.dtype = torch.float16):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.use_cache = use_cache
        self.compute_dtype = compute_dtype
        self.register_buffer('weight_int8', torch.zeros(out_features, in_features, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(out_features, dtype=torch.float32, device=device))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features, dtype=compute_dtype, device=device))
        else:
            self.register_parameter('bias', None)
        self._weight_cache: Optional[Tensor] = None

    def _get_weight(self) -> Tensor:
        """Dequantized weight in compute_dtype, cached when use_cache (reference :70-85).  Kept for
        API compatibility (LoRA merges, debugging); forward() does not need it."""
        if self.use_cache and self._weight_cache is not None:
            return self._weight_cache
        weight = dequantize_rowwise(self.weight_int8, self.weight_scales, dtype=self.compute_dtype)
        if self.use_cache:
            self._weight_cache = weight
        return weight

    def clear_cache(self):
        self._weight_cache = None

    def forward(self, x: Tensor) -> Tensor:
        """x [..., in_features] -> [..., out_features] (reference :91-102), fused W8A16 kernel."""
        return linear_int8(x, self.weight_int8, self.weight_scales, self.bias, dtype=self.compute_dtype)

    @classmethod
    def from_linear(cls, linear: nn.Linear, device=None, use_cache: bool = True,
                    compute_dtype: Optional[torch.dtype] = None) -> 'Linear8bit':
        """Convert an nn.Linear (reference :104-151); weights are quantized on `device` by the HIP kernel."""
        if device is None:
            device = linear.weight.device
        if compute_dtype is None:
            compute_dtype = torch.bfloat16 if linear.weight.dtype == torch.bfloat16 else torch.float16
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=device,
                    use_cache=use_cache, compute_dtype=compute_dtype)
        weight_int8, weight_scales = quantize_rowwise(linear.weight.data.to(device))
        layer.weight_int8.copy_(weight_int8)
        layer.weight_scales.copy_(weight_scales.to(torch.float32))
        if linear.bias is not None:
            layer.bias.data.copy_(linear.bias.data.to(compute_dtype).to(device))
        return layer

    @property
    def device(self) -> torch.device:
        return self.weight_int8.device

    def extra_repr(self) -> str:
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'
