"""
Model-level entry points: BitsAndBytesConfig, replace_linear_with_4bit / _8bit, quantize_model,
get_memory_footprint — the caller of the quantized-linear path.

Same names, fields and defaults as the reference (mps_bitsandbytes/integration.py:16-287), with two
deliberate differences:
  * `device` defaults to 'cuda' (ROCm).  Quantization runs on the GPU, so every nn.Linear weight is
    moved to `device` one layer at a time by `from_linear` and quantized there by the HIP kernel; the
    full-precision model never has to fit on the GPU at once.
  * `bnb_4bit_use_double_quant` is honoured (it becomes `compress_statistics`); the reference accepts the
    flag but does not forward it (integration.py:139-143, docstring :32).
"""
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
from torch import nn

from .nn import Linear4bit, Linear8bit


_dtype_coercion_warned = False


def _warn_dtype_coercion(name: str) -> None:
    global _dtype_coercion_warned
    if not _dtype_coercion_warned:
        _dtype_coercion_warned = True
        import warnings
        warnings.warn(f"BitsAndBytesConfig.from_dict: the compute dtype string {name!r} parses to torch.float16, as in the "
                      "reference (its parser tests 'float16' first, which every dtype name it accepts contains or falls back "
                      "to); pass a torch.dtype to keep bfloat16 / float32", RuntimeWarning, stacklevel=3)


@dataclass
class BitsAndBytesConfig:
    """Quantization config with the field names of transformers' BitsAndBytesConfig
    (reference: integration.py:16-105)."""
    load_in_8bit: bool = False
    load_in_4bit: bool = False
    llm_int8_threshold: float = 6.0
    llm_int8_skip_modules: Optional[list] = None
    llm_int8_enable_fp32_cpu_offload: bool = False
    llm_int8_has_fp16_weight: bool = False
    bnb_4bit_compute_dtype: torch.dtype = torch.float16
    bnb_4bit_quant_type: str = "nf4"
    bnb_4bit_use_double_quant: bool = False
    bnb_4bit_quant_storage: torch.dtype = torch.uint8

    def __post_init__(self):
        if self.load_in_4bit and self.load_in_8bit:
            raise ValueError("Cannot load in both 4-bit and 8-bit")
        if self.bnb_4bit_quant_type not in ('nf4', 'fp4'):
            raise ValueError(f"bnb_4bit_quant_type must be 'nf4' or 'fp4', got {self.bnb_4bit_quant_type}")
        if self.llm_int8_skip_modules is None:
            self.llm_int8_skip_modules = []

    _EXPORTED = ('load_in_8bit', 'load_in_4bit', 'llm_int8_threshold', 'llm_int8_skip_modules',
                 'bnb_4bit_compute_dtype', 'bnb_4bit_quant_type', 'bnb_4bit_use_double_quant')

    def to_dict(self) -> Dict[str, Any]:
        """The serialisable subset (same keys as the reference; the compute dtype as its string name)."""
        out = {name: getattr(self, name) for name in self._EXPORTED}
        out['bnb_4bit_compute_dtype'] = str(out['bnb_4bit_compute_dtype'])
        return out

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> 'BitsAndBytesConfig':
        d = dict(config_dict)
        dt = d.get('bnb_4bit_compute_dtype')
        if isinstance(dt, str):
            # Reproduces the reference exactly (integration.py:86-92): it tests 'float16' first, and 'torch.bfloat16'
            # CONTAINS 'float16' -- so every string, 'torch.bfloat16' and 'float32' included, comes back as torch.float16
            # there.  Kept (a config round trip must build the same layers as the reference does), but never silently: a
            # bf16 / f32 name that is coerced to fp16 raises a RuntimeWarning (once per process).  A torch.dtype value
            # passes through unchanged.  INTEGRATION.md §4.
            if 'float16' not in dt or 'bfloat16' in dt:
                _warn_dtype_coercion(dt)
            d['bnb_4bit_compute_dtype'] = torch.float16
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})

    @property
    def is_quantizable(self) -> bool:
        return self.load_in_4bit or self.load_in_8bit

    @property
    def quantization_method(self) -> str:
        if self.load_in_4bit:
            return 'bitsandbytes_4bit'
        if self.load_in_8bit:
            return 'bitsandbytes_8bit'
        return 'none'


def _walk_and_replace(model: nn.Module, make, skip, prefix):
    for name, module in model.named_children():
        full_name = f"{prefix}.{name}" if prefix else name
        if isinstance(module, nn.Linear):
            if any(s in full_name for s in skip):
                continue
            setattr(model, name, make(module))
        else:
            _walk_and_replace(module, make, skip, full_name)
    return model


def replace_linear_with_4bit(model: nn.Module, quantization_config: BitsAndBytesConfig,
                             modules_to_not_convert: Optional[list] = None,
                             current_key_name: Optional[str] = None, device=None) -> nn.Module:
    """Swap every nn.Linear (except names containing an entry of `modules_to_not_convert`) for a
    Linear4bit quantized on `device` (reference: integration.py:108-154)."""
    skip = modules_to_not_convert or []
    cfg = quantization_config

    def make(linear):
        return Linear4bit.from_linear(linear, device=device, compute_dtype=cfg.bnb_4bit_compute_dtype,
                                      quant_type=cfg.bnb_4bit_quant_type,
                                      compress_statistics=cfg.bnb_4bit_use_double_quant)
    return _walk_and_replace(model, make, skip, current_key_name)


def replace_linear_with_8bit(model: nn.Module, quantization_config: BitsAndBytesConfig,
                             modules_to_not_convert: Optional[list] = None,
                             current_key_name: Optional[str] = None, device=None) -> nn.Module:
    """Swap nn.Linear layers for Linear8bit (reference: integration.py:157-196)."""
    skip = modules_to_not_convert if modules_to_not_convert is not None else (quantization_config.llm_int8_skip_modules or [])
    return _walk_and_replace(model, lambda linear: Linear8bit.from_linear(linear, device=device), skip, current_key_name)


def quantize_model(model: nn.Module, quantization_config: Optional[BitsAndBytesConfig] = None,
                   load_in_4bit: bool = False, load_in_8bit: bool = False, device: str = 'cuda',
                   compute_dtype: torch.dtype = torch.float16,
                   modules_to_not_convert: Optional[list] = None) -> nn.Module:
    """Quantize a model's linear layers and move it to `device` (reference: integration.py:199-251)."""
    if quantization_config is None:
        quantization_config = BitsAndBytesConfig(load_in_4bit=load_in_4bit, load_in_8bit=load_in_8bit,
                                                 bnb_4bit_compute_dtype=compute_dtype)
    if quantization_config.load_in_4bit:
        model = replace_linear_with_4bit(model, quantization_config, modules_to_not_convert, device=device)
    elif quantization_config.load_in_8bit:
        model = replace_linear_with_8bit(model, quantization_config, modules_to_not_convert, device=device)
    return model.to(device)


def get_memory_footprint(model: nn.Module) -> Dict[str, Any]:
    """Parameter/buffer byte counts vs an all-fp16 model (reference: integration.py:254-287).  Unlike the
    reference, whose name filter ('weight_packed'/'weight_int8') misses Linear4bit's `weight` buffer,
    quantized buffers are recognised by owner module type."""
    tensors = [t for _, t in model.named_parameters()] + [t for _, t in model.named_buffers()]
    n_values = sum(t.numel() for t in tensors)
    n_bytes = sum(t.numel() * t.element_size() for t in tensors)
    n_quantized = sum(m.weight.numel() if isinstance(m, Linear4bit) else m.weight_int8.numel()
                      for m in model.modules() if isinstance(m, (Linear4bit, Linear8bit)))
    as_fp16_gb, actual_gb = n_values * 2 / 1e9, n_bytes / 1e9
    return {'total_params': n_values, 'quantized_params': n_quantized, 'fp16_size_gb': as_fp16_gb,
            'actual_size_gb': actual_gb, 'savings_gb': as_fp16_gb - actual_gb,
            'savings_pct': (1 - actual_gb / as_fp16_gb) * 100 if as_fp16_gb > 0 else 0}
