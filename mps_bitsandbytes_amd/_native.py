"""
ctypes binding of the gfx950 kernel library (include/mbnb_hip.h).

This is the counterpart of the reference's ``_try_load_native()`` / ``from . import _C``
(functional.py:50-56, __init__.py:125-131) — with one deliberate difference: there is NO
Python/CPU fallback behind it.  If ``libmbnb_hip.so`` is missing or a call fails, the
caller gets a RuntimeError; nothing silently degrades.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmbnb_hip.so")

ABI_VERSION = 2   # include/mbnb_hip.h MBNB_ABI_VERSION
F16, BF16, F32 = 0, 1, 2
NF4, FP4 = 0, 1
DTYPE_CODE = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}
QUANT_CODE = {"nf4": NF4, "fp4": FP4}


class AbsmaxDesc(Structure):
    """mirror of ``struct mbnb_absmax`` (include/mbnb_hip.h)"""
    _fields_ = [("absmax_f32", c_void_p), ("absmax_i8", c_void_p), ("absmax2", c_void_p),
                ("blocksize2", c_int32)]


_lib = None
_load_error: Optional[str] = None

_SIGNATURES = {
    "mbnb_abi_version": (c_int, []),
    "mbnb_last_error": (c_char_p, []),
    "mbnb_last_kernel": (c_char_p, []),
    "mbnb_quantize_4bit": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "mbnb_quantize_4bit_dq": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "mbnb_dequantize_4bit": (c_int, [c_void_p, POINTER(AbsmaxDesc), c_int64, c_int64, c_int64, c_int, c_int,
                                     c_int, c_void_p, c_void_p]),
    "mbnb_quantize_blockwise": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mbnb_dequantize_blockwise": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mbnb_dequant_absmax": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mbnb_quantize_rowwise": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mbnb_dequantize_rowwise": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mbnb_double_quant": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_void_p]),
    "mbnb_matmul_4bit_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int64, c_int, c_int]),
    "mbnb_matmul_4bit": (c_int, [c_void_p, c_int64, c_int64, c_void_p, POINTER(AbsmaxDesc), c_int64, c_int64,
                                 c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mbnb_matmul_int8_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "mbnb_matmul_int8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                 c_void_p, c_void_p, c_int64, c_void_p]),
    "mbnb_gemm_dense_applies": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "mbnb_gemm_dense_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "mbnb_gemm_dense": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
                        c_int64, c_int, c_void_p]),
    "mbnb_linear_int8_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int]),
    "mbnb_linear_int8": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                 c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mbnb_quantize_fp8_e4m3": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mbnb_dequantize_fp8_e4m3": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mbnb_linear_fp8": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                c_void_p, c_int64, c_int, c_void_p]),
    "mbnb_embedding_4bit": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                    c_int64, c_int, c_void_p, c_void_p]),
    "mbnb_embedding_8bit": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_int,
                                    c_void_p, c_void_p]),
    "mbnb_outlier_linear_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "mbnb_outlier_linear": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def available() -> bool:
    """True when libmbnb_hip.so is present and loads (does not need a GPU)."""
    try:
        lib()
        return True
    except RuntimeError:
        return False


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it cannot be loaded."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise RuntimeError(_load_error)
    if not os.path.exists(LIB_PATH):
        _load_error = (f"mps_bitsandbytes_amd: native library {LIB_PATH} not found. Build it with "
                       f"`make -C {os.path.join(_HERE, 'csrc')}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
                       f"There is no Python fallback.")
        raise RuntimeError(_load_error)
    try:
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.mbnb_abi_version() != ABI_VERSION:
            raise OSError(f"ABI version mismatch: library reports {handle.mbnb_abi_version()}, binding expects {ABI_VERSION}")
    except (OSError, AttributeError) as e:
        _load_error = f"mps_bitsandbytes_amd: cannot load {LIB_PATH}: {e}"
        raise RuntimeError(_load_error) from e
    _lib = handle
    return _lib


def last_kernel() -> str:
    return lib().mbnb_last_kernel().decode()


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mbnb_last_error().decode(errors="replace")
        raise RuntimeError(f"mps_bitsandbytes_amd.{what} failed (status {status}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> c_void_p:
    """The caller's current HIP stream on `device` as the ABI's `void *stream`.  torch's raw-stream query when this build has it (no Stream object
    is made: 0.3 us instead of 4 us of a 15 us eager M = 1 call, tools/host_overhead.py)."""
    if not isinstance(device, torch.device):
        device = torch.device(device)       # "cuda", "cuda:1", 0 ...
    if _raw_stream is not None:
        idx = device.index
        return c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device: torch.device):
    """`with on_device(t.device):` -- torch.cuda.device(...) only when `device` is not already the current one (the guard object and its two
    device switches cost 2.5 us per call; a single-GPU process never needs them)."""
    if not isinstance(device, torch.device):
        device = torch.device(device)
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def dtype_code(dtype: torch.dtype, what: str) -> int:
    try:
        return DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError(f"mps_bitsandbytes_amd {what}: unsupported dtype {dtype} "
                        f"(supported: float16, bfloat16, float32)") from None
