"""
Outlier-aware INT8 linear, LLM.int8()-style (reference: mps_bitsandbytes/nn/outlier_aware.py).

Same class, arguments, buffers (`weight_int8`, `weight_scales`, `outlier_indices`, `outlier_weights`, `bias`) and
`from_linear` analysis as the reference.  The forward is `functional.outlier_linear` (C ABI `mbnb_outlier_linear`):
masked row-wise quantisation, the int8 x int8 MFMA GEMM over the stored weight, and one small kernel for the
outlier columns + bias.  ROCm (`cuda`) tensors only.
"""
import torch
from torch import nn, Tensor

from .. import functional as F


class OutlierAwareLinear(nn.Module):
    """INT8 linear with mixed-precision outlier decomposition (reference: nn/outlier_aware.py:18-219)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, threshold: float = 6.0,
                 compute_dtype=torch.float16, device=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.threshold = threshold
        self.compute_dtype = compute_dtype
        self.register_buffer('weight_int8', torch.zeros(out_features, in_features, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(out_features, dtype=torch.float32, device=device))
        self.register_buffer('outlier_indices', torch.tensor([], dtype=torch.long, device=device))
        self.register_buffer('outlier_weights', torch.zeros(out_features, 0, dtype=compute_dtype, device=device))
        if bias:
            self.register_buffer('bias', torch.zeros(out_features, dtype=compute_dtype, device=device))
        else:
            self.register_buffer('bias', None)

    def forward(self, x: Tensor) -> Tensor:
        """x [..., in_features] -> [..., out_features] in compute_dtype (reference: nn/outlier_aware.py:84-113)."""
        return F.outlier_linear(x, self.weight_int8, self.weight_scales, self.outlier_indices, self.outlier_weights,
                                self.bias, self.compute_dtype)

    @classmethod
    def from_linear(cls, linear: nn.Linear, threshold: float = 6.0, device=None) -> 'OutlierAwareLinear':
        """Reference: nn/outlier_aware.py:148-212.  Outlier columns are those whose largest |weight| exceeds
        threshold x mean|weight|; they are kept in compute_dtype and zeroed before the row-wise INT8 quantiser."""
        if device is None:
            device = linear.weight.device
        dtype = linear.weight.dtype
        if dtype not in (torch.float16, torch.bfloat16):
            dtype = torch.float16
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, threshold=threshold,
                    compute_dtype=dtype, device=device)
        weight = linear.weight.data.to(device)
        col_max = weight.abs().max(dim=0).values
        mean_abs = weight.abs().mean()
        outlier_indices = torch.where(col_max > (threshold * mean_abs))[0]
        if len(outlier_indices) > 0:
            layer.outlier_indices = outlier_indices
            layer.outlier_weights = weight[:, outlier_indices].to(dtype)
            weight_for_int8 = weight.clone()
            weight_for_int8[:, outlier_indices] = 0
        else:
            weight_for_int8 = weight
        weight_int8, weight_scales = F.quantize_rowwise(weight_for_int8)
        layer.weight_int8.copy_(weight_int8)
        layer.weight_scales.copy_(weight_scales)
        if linear.bias is not None:
            layer.bias.copy_(linear.bias.data.to(dtype))
        return layer

    def extra_repr(self) -> str:
        return (f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, '
                f'threshold={self.threshold}, outliers={len(self.outlier_indices)}')
