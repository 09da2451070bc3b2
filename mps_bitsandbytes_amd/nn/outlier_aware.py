"""
OutlierAwareLinear — LLM.int8()-style INT8 linear with the outlier input columns kept in 16 bit, on MI355X.

Public surface of the reference module (mps_bitsandbytes/nn/outlier_aware.py:18-219): constructor arguments, buffers
(`weight_int8`, `weight_scales`, `outlier_indices`, `outlier_weights`, `bias`) and the `from_linear` outlier analysis.
The forward is `functional.outlier_linear`: masked row-wise quantisation of the activations, the int8 x int8 MFMA GEMM
over the stored weight, one small kernel for the outlier columns + bias.
"""
import torch
from torch import nn, Tensor

from .. import functional as F
from ._base import QuantizedModule, sixteen_bit_or_half, source_device


class OutlierAwareLinear(QuantizedModule):
    _anchor = 'weight_int8'

    def __init__(self, in_features: int, out_features: int, bias: bool = True, threshold: float = 6.0,
                 compute_dtype=torch.float16, device=None):
        super().__init__()
        self._init_linear(in_features, out_features, compute_dtype)
        self.threshold = threshold
        self.register_buffer('weight_int8', torch.zeros(out_features, in_features, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(out_features, dtype=torch.float32, device=device))
        # the outlier input columns and their weights [out_features, n_outliers]; empty until from_linear finds some
        self.register_buffer('outlier_indices', torch.empty(0, dtype=torch.long, device=device))
        self.register_buffer('outlier_weights', torch.zeros(out_features, 0, dtype=compute_dtype, device=device))
        self._init_bias(bias, device, as_parameter=False)   # a buffer, as in the reference

    def forward(self, x: Tensor) -> Tensor:
        """[..., in_features] -> [..., out_features] in compute_dtype (reference :84-146)."""
        return F.outlier_linear(x, self.weight_int8, self.weight_scales, self.outlier_indices, self.outlier_weights,
                                self.bias, self.compute_dtype)

    @classmethod
    def from_linear(cls, linear: nn.Linear, threshold: float = 6.0, device=None) -> 'OutlierAwareLinear':
        """Reference :148-212.  An input column is an outlier when its largest |weight| exceeds threshold x mean|weight|;
        outlier columns stay in compute_dtype and are zeroed before the row-wise INT8 quantiser."""
        device = source_device(linear.weight, device)
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, threshold=threshold,
                    compute_dtype=sixteen_bit_or_half(linear.weight.dtype), device=device)
        w = linear.weight.data.to(device)
        magnitude = w.abs()
        columns = torch.where(magnitude.max(dim=0).values > threshold * magnitude.mean())[0]
        if columns.numel():
            layer.outlier_indices = columns
            layer.outlier_weights = w[:, columns].to(layer.compute_dtype)
            w = w.clone()
            w[:, columns] = 0
        q, absmax = F.quantize_rowwise(w)
        layer.weight_int8.copy_(q)
        layer.weight_scales.copy_(absmax)
        layer._copy_bias_from(linear, device)
        return layer

    def extra_repr(self) -> str:
        return f'{self._repr_core()}, threshold={self.threshold}, outliers={len(self.outlier_indices)}'
