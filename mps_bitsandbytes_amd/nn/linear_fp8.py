"""
LinearFP8 — FP8 E4M3 (the reference's format) row-wise quantized linear layer on MI355X.

Public surface of the reference module (mps_bitsandbytes/nn/linear_fp8.py:17-168): buffers `weight_fp8` [N, K] uint8 and
`weight_scales` [N] f32, `bias` Parameter, `from_linear`, `dequantize`.  The forward is `functional.matmul_fp8_e4m3`:
the W8A16 GEMM kernels with the FP8 byte decoder in the weight producer.
"""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F
from ._base import QuantizedModule, compute_dtype_for, source_device


class LinearFP8(QuantizedModule):
    _anchor = 'weight_fp8'

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 compute_dtype: torch.dtype = torch.float16):
        super().__init__()
        self._init_linear(in_features, out_features, compute_dtype)
        self.register_buffer('weight_fp8', torch.zeros(out_features, in_features, dtype=torch.uint8, device=device))
        self.register_buffer('weight_scales', torch.ones(out_features, dtype=torch.float32, device=device))
        self._init_bias(bias, device)

    def forward(self, x: Tensor) -> Tensor:
        return F.matmul_fp8_e4m3(x, self.weight_fp8, self.weight_scales, self.bias, self.compute_dtype)

    @classmethod
    def from_linear(cls, linear: nn.Linear, device=None, compute_dtype: Optional[torch.dtype] = None) -> 'LinearFP8':
        """Quantize an nn.Linear on `device` with the HIP kernel (reference :105-151)."""
        device = source_device(linear.weight, device)
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=device,
                    compute_dtype=compute_dtype_for(linear.weight.dtype, compute_dtype))
        q, scales = F.quantize_fp8_e4m3(linear.weight.data.to(device))
        layer.weight_fp8.copy_(q)
        layer.weight_scales.copy_(scales)
        layer._copy_bias_from(linear, device)
        return layer

    def dequantize(self) -> Tensor:
        """The weight as [out_features, in_features] in compute_dtype (reference :153-163)."""
        return F.dequantize_fp8_e4m3(self.weight_fp8, self.weight_scales, self.compute_dtype)

    def extra_repr(self) -> str:
        return f'{self._repr_core()}, quant_type=fp8_e4m3'
