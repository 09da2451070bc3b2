"""
Shared plumbing of the quantized layers (no counterpart in the reference, which repeats these pieces per module).

`QuantizedModule` keeps what every layer here needs — feature sizes, the optional bias in the compute dtype, the
device the quantized buffers live on, leading-dimension handling — so that the layer files only state their
storage format and which HIP entry point runs their forward.
"""
from typing import Optional, Tuple

import torch
from torch import nn, Tensor

_SIXTEEN_BIT = (torch.float16, torch.bfloat16)


def compute_dtype_for(weight_dtype: torch.dtype, requested: Optional[torch.dtype] = None) -> torch.dtype:
    """bf16 weights compute in bf16, everything else in fp16, unless the caller asked for a dtype."""
    if requested is not None:
        return requested
    return torch.bfloat16 if weight_dtype == torch.bfloat16 else torch.float16


def sixteen_bit_or_half(dtype: torch.dtype) -> torch.dtype:
    """Embedding / outlier layers keep a 16-bit source dtype and fall back to fp16 otherwise."""
    return dtype if dtype in _SIXTEEN_BIT else torch.float16


def source_device(tensor: Tensor, device) -> torch.device:
    return tensor.device if device is None else torch.device(device)


def fold_leading(x: Tensor, width: int) -> Tuple[Tensor, Tuple[int, ...]]:
    """[..., width] -> ([rows, width], leading shape)."""
    lead = tuple(x.shape[:-1])
    return (x if x.dim() == 2 else x.reshape(-1, width)), lead


class QuantizedModule(nn.Module):
    """Base of Linear4bit / Linear8bit / OutlierAwareLinear: sizes, bias, device, repr."""

    #: name of the buffer whose device is "the layer's device"
    _anchor = None

    def _init_linear(self, in_features: int, out_features: int, compute_dtype: torch.dtype) -> None:
        self.in_features, self.out_features, self.compute_dtype = in_features, out_features, compute_dtype

    def _init_bias(self, enabled: bool, device, as_parameter: bool = True) -> None:
        if not enabled:
            if as_parameter:
                self.register_parameter('bias', None)
            else:
                self.register_buffer('bias', None)
            return
        zeros = torch.zeros(self.out_features, dtype=self.compute_dtype, device=device)
        if as_parameter:
            self.bias = nn.Parameter(zeros)
        else:
            self.register_buffer('bias', zeros)

    def _copy_bias_from(self, linear: nn.Linear, device) -> None:
        if linear.bias is not None:
            target = self.bias.data if isinstance(self.bias, nn.Parameter) else self.bias
            target.copy_(linear.bias.data.to(self.compute_dtype).to(device))

    @property
    def device(self) -> torch.device:
        anchor = getattr(self, self._anchor, None) if self._anchor else None
        if anchor is not None and anchor.numel() > 0:
            return anchor.device
        bias = getattr(self, 'bias', None)
        return bias.device if bias is not None else torch.device('cpu')

    def _repr_core(self) -> str:
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'
