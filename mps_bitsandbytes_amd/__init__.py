"""
mps_bitsandbytes_amd — MI355X (gfx950) native quantized-linear backend with the
bitsandbytes-style API of mpsops/mps-bitsandbytes (reference: mps_bitsandbytes/__init__.py).

Scope: the quantized-linear hot path — NF4/FP4 blockwise 4-bit and rowwise INT8:
quantize / dequantize / fused dequant+matmul, `QuantState`, `Linear4bit`, `Linear8bit`.
Everything runs as hand-written HIP kernels behind a C ABI (include/mbnb_hip.h); there is
no CPU fallback.
"""
import torch as _torch

__version__ = "0.1.0"

from .functional import (
    QuantState,
    quantize_4bit, dequantize_4bit, matmul_4bit,
    quantize_nf4, dequantize_nf4, matmul_nf4, NF4_CODEBOOK, create_normal_map,
    quantize_fp4, dequantize_fp4, matmul_fp4, FP4_CODEBOOK, create_fp4_map,
    quantize_blockwise, dequantize_blockwise,
    quantize_rowwise, dequantize_rowwise, matmul_int8, linear_int8,
    double_quant, dequant_absmax, embedding_4bit, embedding_8bit, outlier_linear,
    quantize_fp8_e4m3, dequantize_fp8_e4m3, matmul_fp8_e4m3,
)
from .nn import (Linear4bit, Linear8bit, LinearFP8, Params4bit, Embedding4bit, Embedding8bit, EmbeddingNF4, EmbeddingFP4,
                 OutlierAwareLinear)
from .integration import (
    BitsAndBytesConfig, quantize_model, replace_linear_with_4bit, replace_linear_with_8bit, get_memory_footprint,
)


def is_available() -> bool:
    """True when a ROCm GPU is visible and the native kernel library loads
    (reference: is_available(), __init__.py:120-122, which checks MPS)."""
    return bool(_torch.cuda.is_available()) and has_native_kernels()


def has_native_kernels() -> bool:
    """True when libmbnb_hip.so is built and loadable (reference: __init__.py:125-131)."""
    from . import _native
    return _native.available()


__all__ = [
    '__version__', 'is_available', 'has_native_kernels', 'QuantState',
    'quantize_4bit', 'dequantize_4bit', 'matmul_4bit',
    'quantize_nf4', 'dequantize_nf4', 'matmul_nf4', 'NF4_CODEBOOK', 'create_normal_map',
    'quantize_fp4', 'dequantize_fp4', 'matmul_fp4', 'FP4_CODEBOOK', 'create_fp4_map',
    'quantize_blockwise', 'dequantize_blockwise',
    'quantize_rowwise', 'dequantize_rowwise', 'matmul_int8', 'linear_int8',
    'double_quant', 'dequant_absmax',
    'Linear4bit', 'Linear8bit', 'Params4bit', 'Embedding4bit', 'Embedding8bit', 'EmbeddingNF4', 'EmbeddingFP4',
    'OutlierAwareLinear', 'embedding_4bit', 'embedding_8bit', 'outlier_linear',
    'LinearFP8', 'quantize_fp8_e4m3', 'dequantize_fp8_e4m3', 'matmul_fp8_e4m3',
    'BitsAndBytesConfig', 'quantize_model', 'replace_linear_with_4bit', 'replace_linear_with_8bit', 'get_memory_footprint',
]
