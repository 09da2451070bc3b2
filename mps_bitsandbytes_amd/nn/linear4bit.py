"""
Linear4bit — 4-bit NF4/FP4 quantized linear layer on MI355X.

Same constructor, attributes, state-dict keys and hooks as the reference module
(mps_bitsandbytes/nn/linear4bit.py:18-337); `forward` is one fused HIP kernel
(functional.matmul_4bit) for every batch size.
"""
import warnings
from typing import Optional

import torch
from torch import nn, Tensor

from ..functional import quantize_4bit, dequantize_4bit, matmul_4bit, QuantState


class Linear4bit(nn.Module):
    """
    4-bit quantized linear layer (NF4 or FP4 blockwise absmax, optional double quantization).

    Storage (reference nn/linear4bit.py:26-28, :76-91): ``weight`` — packed uint8 buffer (flat);
    ``weight_quant_state`` — QuantState (absmax, shape, blocksize, quant_type, dtype, state2);
    ``bias`` — nn.Parameter in ``compute_dtype`` or None.
    """

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 compute_dtype: torch.dtype = torch.float16, quant_type: str = 'nf4', blocksize: int = 64,
                 compress_statistics: bool = False):
        super().__init__()
        if quant_type not in ('nf4', 'fp4'):
            raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
        self.in_features = in_features
        self.out_features = out_features
        self.compute_dtype = compute_dtype
        self.quant_type = quant_type
        self.blocksize = blocksize
        self.compress_statistics = compress_statistics

        # placeholder sized from the flat numel, as the reference does (nn/linear4bit.py:70-80);
        # from_linear / load_state_dict replace it with the row-padded size
        numel = out_features * in_features
        padded_numel = ((numel + blocksize - 1) // blocksize) * blocksize
        if padded_numel % 2 != 0:
            padded_numel += blocksize
        self.register_buffer('weight', torch.zeros(padded_numel // 2, dtype=torch.uint8, device=device))
        self.weight_quant_state: Optional[QuantState] = None
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features, dtype=compute_dtype, device=device))
        else:
            self.register_parameter('bias', None)

    def forward(self, x: Tensor) -> Tensor:
        """x [..., in_features] -> [..., out_features]; fused dequant + matmul (nn/linear4bit.py:93-119)."""
        if self.weight_quant_state is None:
            raise RuntimeError("Weight not quantized. Call from_linear() or load weights first.")
        orig_shape = x.shape
        if x.dim() > 2:
            x = x.reshape(-1, self.in_features)
        output = matmul_4bit(x, self.weight, self.weight_quant_state, self.bias, compute_dtype=self.compute_dtype)
        if len(orig_shape) > 2:
            output = output.reshape(*orig_shape[:-1], self.out_features)
        return output

    @classmethod
    def from_linear(cls, linear: nn.Linear, device=None, compute_dtype: Optional[torch.dtype] = None,
                    quant_type: str = 'nf4', blocksize: int = 64,
                    compress_statistics: bool = False) -> 'Linear4bit':
        """Convert an nn.Linear (reference nn/linear4bit.py:121-190).  The weight is moved to
        `device` (default: where it already is) and quantized there by the HIP kernel."""
        if device is None:
            device = linear.weight.device
        if compute_dtype is None:
            compute_dtype = torch.bfloat16 if linear.weight.dtype == torch.bfloat16 else torch.float16
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=device,
                    compute_dtype=compute_dtype, quant_type=quant_type, blocksize=blocksize,
                    compress_statistics=compress_statistics)
        weight = linear.weight.data.to(device)
        weight_packed, quant_state = quantize_4bit(weight, blocksize=blocksize,
                                                   compress_statistics=compress_statistics, quant_type=quant_type)
        layer.weight = layer.weight.new_zeros(weight_packed.numel())
        layer.weight.copy_(weight_packed)
        layer.weight_quant_state = quant_state
        if linear.bias is not None:
            layer.bias.data.copy_(linear.bias.data.to(compute_dtype).to(device))
        return layer

    def dequantize(self) -> Tensor:
        """Dequantized weight [out_features, in_features] (reference nn/linear4bit.py:192-204)."""
        if self.weight_quant_state is None:
            raise RuntimeError("Weight not quantized")
        return dequantize_4bit(self.weight, self.weight_quant_state)

    @property
    def quant_state(self):
        return self.weight_quant_state

    @property
    def device(self) -> torch.device:
        if self.weight is not None and self.weight.numel() > 0:
            return self.weight.device
        if self.bias is not None:
            return self.bias.device
        return torch.device('cpu')

    def _apply(self, fn):
        """Move the QuantState along with the buffers on .to()/.cuda() (reference :230-236)."""
        super()._apply(fn)
        if self.weight_quant_state is not None:
            self.weight_quant_state.to(self.weight.device)
        return self

    def extra_repr(self) -> str:
        return (f'in_features={self.in_features}, out_features={self.out_features}, '
                f'bias={self.bias is not None}, quant_type={self.quant_type}, blocksize={self.blocksize}')

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.weight_quant_state is not None:
            destination[prefix + 'weight_quant_state'] = self.weight_quant_state.as_dict()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """Checkpoint contract of the reference (nn/linear4bit.py:251-312): adopt the checkpoint's
        blocksize / quant_type with a warning, rebuild the QuantState on this layer's device, and
        quantize on the fly when `weight` arrives in fp16/bf16/fp32."""
        target_device = self.weight.device
        quant_state_key = prefix + 'weight_quant_state'
        if quant_state_key in state_dict:
            loaded_state = state_dict.pop(quant_state_key)
            loaded_blocksize = loaded_state.get('blocksize', 64)
            if loaded_blocksize != self.blocksize:
                warnings.warn(
                    f"Linear4bit blocksize mismatch: layer has blocksize={self.blocksize}, "
                    f"checkpoint has blocksize={loaded_blocksize}. Using checkpoint blocksize.", UserWarning)
                self.blocksize = loaded_blocksize
            loaded_quant_type = loaded_state.get('quant_type', 'nf4')
            if loaded_quant_type != self.quant_type:
                warnings.warn(
                    f"Linear4bit quant_type mismatch: layer has quant_type='{self.quant_type}', "
                    f"checkpoint has quant_type='{loaded_quant_type}'. Using checkpoint quant_type.", UserWarning)
                self.quant_type = loaded_quant_type
            self.weight_quant_state = QuantState.from_dict(loaded_state, device=target_device)

        weight_key = prefix + 'weight'
        if weight_key in state_dict:
            weight_data = state_dict[weight_key]
            if weight_data.device != target_device:
                weight_data = weight_data.to(target_device)
            if weight_data.dtype in (torch.float16, torch.float32, torch.bfloat16):
                weight_packed, quant_state = quantize_4bit(weight_data, blocksize=self.blocksize,
                                                           compress_statistics=self.compress_statistics,
                                                           quant_type=self.quant_type)
                weight_data = weight_packed
                self.weight_quant_state = quant_state
            state_dict[weight_key] = weight_data
            if weight_data.numel() != self.weight.numel():
                # the constructor sizes the buffer from the flat numel, a checkpoint from the
                # row-padded one; adopt the checkpoint's size
                self.weight = torch.zeros(weight_data.numel(), dtype=torch.uint8, device=target_device)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class Params4bit(nn.Parameter):
    """Parameter wrapper exposing the logical (unpacked) shape (reference nn/linear4bit.py:315-337)."""

    def __new__(cls, data=None, requires_grad=False, quant_state=None):
        if data is None:
            data = torch.empty(0)
        instance = torch.Tensor._make_subclass(cls, data, requires_grad)
        instance.quant_state = quant_state
        return instance

    @property
    def shape(self):
        if hasattr(self, 'quant_state') and self.quant_state is not None:
            if isinstance(self.quant_state, QuantState):
                return self.quant_state.shape
            return self.quant_state.get('shape', super().shape)
        return super().shape
