"""
Linear8bit — row-wise INT8 linear layer on MI355X.

Public surface of the reference module (mps_bitsandbytes/nn/linear8bit.py:15-166): buffers `weight_int8` [N, K] int8
and `weight_scales` [N] f32 (the row absmax), `from_linear`, the dequantized-weight cache API and `device`.
Forward: `functional.linear_int8` decodes the int8 rows inside the GEMM at small and mid-sized batches (skinny MFMA,
`k_gemm_small8`, split-K).  At large batches (from 256 rows / 1.5 M outputs) the kernels are the reference's own two steps,
dequantize_rowwise -> dense GEMM; with `use_cache` (the default, as in the reference) the dequantised weight of that step is
KEPT in `_weight_cache` (reference :70-85) and later large-batch calls run the dense GEMM alone -- same bits as the
uncached call, minus the dequantise pass.  `use_cache=False` re-dequantises into transient scratch on every call.

Cost and validity of the cache: it holds N x K x 2 bytes per layer after the layer's first large-batch forward (an int8 model's
weight memory roughly triples once every layer has seen a prefill; `clear_cache()` or `use_cache=False` gives it back).  It is
keyed on the buffers it was made from -- their storage pointers and in-place version counters -- so `load_state_dict`, `copy_`
into `weight_int8` / `weight_scales`, `.to()` / `_apply` all invalidate it (the reference's cache has no such check: there a
reloaded checkpoint keeps serving the old weights until `clear_cache()`).
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
        self._weight_cache_key = None

    def _cache_key(self):
        w, s = self.weight_int8, self.weight_scales
        return (w.data_ptr(), w._version, s.data_ptr(), s._version, w.device, self.compute_dtype)

    def forward(self, x: Tensor) -> Tensor:
        if self.use_cache and self.compute_dtype in (torch.float16, torch.bfloat16) and x.is_cuda:
            K = self.weight_int8.shape[1]
            M = x.numel() // K if K else 0
            if F.DECODE_ONCE and F.dense_path_applies(M, self.weight_int8.shape[0], K):
                return F.linear_dense(x, self._get_weight(), self.bias)
        return F.linear_int8(x, self.weight_int8, self.weight_scales, self.bias, dtype=self.compute_dtype)

    # dequantized weight, kept while use_cache is set (reference :70-89): large-batch forwards, LoRA merges, debugging
    def _get_weight(self) -> Tensor:
        cached = self._weight_cache if self.use_cache else None
        key = self._cache_key()
        if cached is not None and self._weight_cache_key != key:
            cached = None     # the buffers were reloaded, written in place, moved or re-typed since the weight was cached
        if cached is None:
            cached = F.dequantize_rowwise(self.weight_int8, self.weight_scales, dtype=self.compute_dtype)
            if self.use_cache:
                self._weight_cache, self._weight_cache_key = cached, key
            else:
                self._weight_cache = self._weight_cache_key = None
        return cached

    def clear_cache(self):
        self._weight_cache = None
        self._weight_cache_key = None

    def _apply(self, fn, *args, **kwargs):
        self.clear_cache()       # .to() / .cuda() / .half(): the cached copy belongs to the old placement
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.clear_cache()       # a loaded checkpoint replaces the weights the cache was made from
        return super()._load_from_state_dict(*args, **kwargs)

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
