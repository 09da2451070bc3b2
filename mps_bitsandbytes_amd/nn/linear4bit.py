"""
Linear4bit — NF4 / FP4 blockwise-quantized linear layer on MI355X.

Public surface of the reference module (mps_bitsandbytes/nn/linear4bit.py:18-337): constructor arguments,
`from_linear`, `dequantize`, `quant_state`, `device`, state-dict keys (`weight`, `bias`, `weight_quant_state`) and the
quantize-on-load / adopt-the-checkpoint's-format rules.  The forward is `functional.matmul_4bit` for every batch size
(GEMV, skinny MFMA, split-K or the 256 x 256 fused kernel — see DESIGN.md §5).
"""
import warnings
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F
from ..functional import QuantState
from ._base import QuantizedModule, compute_dtype_for, fold_leading, source_device


def _packed_bytes(numel: int, blocksize: int) -> int:
    """Bytes of a flat 4-bit buffer for `numel` values padded to whole blocks (and to a whole byte)."""
    padded = -(-numel // blocksize) * blocksize
    if padded & 1:
        padded += blocksize
    return padded // 2


class Linear4bit(QuantizedModule):
    """`weight`: flat packed uint8 buffer; `weight_quant_state`: QuantState; `bias`: Parameter in compute_dtype or None
    (reference nn/linear4bit.py:26-28, :76-91)."""

    _anchor = 'weight'

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 compute_dtype: torch.dtype = torch.float16, quant_type: str = 'nf4', blocksize: int = 64,
                 compress_statistics: bool = False):
        super().__init__()
        if quant_type not in ('nf4', 'fp4'):
            raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
        self._init_linear(in_features, out_features, compute_dtype)
        self.quant_type, self.blocksize, self.compress_statistics = quant_type, blocksize, compress_statistics
        # sized from the flat element count like the reference's constructor (:70-80); a quantized weight
        # (from_linear / load_state_dict) brings its own, row-padded size
        self.register_buffer('weight', torch.zeros(_packed_bytes(in_features * out_features, blocksize),
                                                   dtype=torch.uint8, device=device))
        self.weight_quant_state: Optional[QuantState] = None
        self._init_bias(bias, device)

    # ------------------------------------------------------------------ compute
    def forward(self, x: Tensor) -> Tensor:
        state = self.weight_quant_state
        if state is None:
            raise RuntimeError("Weight not quantized. Call from_linear() or load weights first.")
        rows, lead = fold_leading(x, self.in_features)
        y = F.matmul_4bit(rows, self.weight, state, self.bias, compute_dtype=self.compute_dtype)
        return y if x.dim() <= 2 else y.reshape(*lead, self.out_features)

    def dequantize(self) -> Tensor:
        """The weight as [out_features, in_features] in the QuantState's dtype (reference :192-204)."""
        if self.weight_quant_state is None:
            raise RuntimeError("Weight not quantized")
        return F.dequantize_4bit(self.weight, self.weight_quant_state)

    # ------------------------------------------------------------------ construction
    def _set_quantized(self, packed: Tensor, state: QuantState) -> None:
        if packed.numel() != self.weight.numel():
            self.weight = torch.zeros(packed.numel(), dtype=torch.uint8, device=packed.device)
        self.weight.copy_(packed)
        self.weight_quant_state = state

    @classmethod
    def from_linear(cls, linear: nn.Linear, device=None, compute_dtype: Optional[torch.dtype] = None,
                    quant_type: str = 'nf4', blocksize: int = 64,
                    compress_statistics: bool = False) -> 'Linear4bit':
        """Quantize an nn.Linear on `device` (default: where its weight is) with the HIP kernel (reference :121-190)."""
        device = source_device(linear.weight, device)
        layer = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=device,
                    compute_dtype=compute_dtype_for(linear.weight.dtype, compute_dtype), quant_type=quant_type,
                    blocksize=blocksize, compress_statistics=compress_statistics)
        layer._set_quantized(*F.quantize_4bit(linear.weight.data.to(device), blocksize=blocksize,
                                              compress_statistics=compress_statistics, quant_type=quant_type))
        layer._copy_bias_from(linear, device)
        return layer

    # ------------------------------------------------------------------ introspection
    @property
    def quant_state(self):
        return self.weight_quant_state

    def extra_repr(self) -> str:
        return f'{self._repr_core()}, quant_type={self.quant_type}, blocksize={self.blocksize}'

    def _apply(self, fn):
        super()._apply(fn)   # buffers first; the QuantState's tensors follow them (reference :230-236)
        if self.weight_quant_state is not None:
            self.weight_quant_state.to(self.weight.device)
        return self

    # ------------------------------------------------------------------ checkpoints (reference :245-312)
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.weight_quant_state is not None:
            destination[prefix + 'weight_quant_state'] = self.weight_quant_state.as_dict()

    def _adopt(self, field: str, loaded, default) -> None:
        """A checkpoint's blocksize / quant_type wins over the layer's, with a warning."""
        value = loaded.get(field, default)
        current = getattr(self, field)
        if value != current:
            shown = (lambda v: f"'{v}'") if field == 'quant_type' else str
            warnings.warn(f"Linear4bit {field} mismatch: layer has {field}={shown(current)}, "
                          f"checkpoint has {field}={shown(value)}. Using checkpoint {field}.", UserWarning)
            setattr(self, field, value)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        here = self.weight.device
        packed_state = state_dict.pop(prefix + 'weight_quant_state', None)
        if packed_state is not None:
            self._adopt('blocksize', packed_state, 64)
            self._adopt('quant_type', packed_state, 'nf4')
            self.weight_quant_state = QuantState.from_dict(packed_state, device=here)
        key = prefix + 'weight'
        incoming = state_dict.get(key)
        if incoming is not None:
            incoming = incoming.to(here)
            if incoming.dtype in (torch.float16, torch.bfloat16, torch.float32):   # unquantized checkpoint: quantize on load
                incoming, self.weight_quant_state = F.quantize_4bit(
                    incoming, blocksize=self.blocksize, compress_statistics=self.compress_statistics,
                    quant_type=self.quant_type)
            if incoming.numel() != self.weight.numel():   # constructor size (flat) vs checkpoint size (row-padded)
                self.weight = torch.zeros(incoming.numel(), dtype=torch.uint8, device=here)
            state_dict[key] = incoming
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class Params4bit(nn.Parameter):
    """Parameter over packed 4-bit data that reports the logical weight shape (reference :315-337)."""

    def __new__(cls, data=None, requires_grad=False, quant_state=None):
        self = torch.Tensor._make_subclass(cls, torch.empty(0) if data is None else data, requires_grad)
        self.quant_state = quant_state
        return self

    @property
    def shape(self):
        state = getattr(self, 'quant_state', None)
        if isinstance(state, QuantState):
            return state.shape
        if state is not None:
            return state.get('shape', super().shape)
        return super().shape
