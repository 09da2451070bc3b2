"""
oracle — ctypes front-end of the CPU restatement in oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg.  The product package never imports it.

All functions take and return CPU ``torch.Tensor``s so the parity tests read
like the reference's own tests; they mirror the reference's signatures
(functional.py) closely but return plain tuples instead of QuantState so this
module has no dependency on the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
_QT = {"nf4": 0, "fp4": 1}


def build(force: bool = False) -> str:
    """Compile liboracle.so with the Makefile next to this file (gcc, no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in ("orc_quantize_4bit", "orc_dequantize_4bit", "orc_quantize_blockwise",
                     "orc_dequantize_blockwise", "orc_dequant_absmax", "orc_quantize_rowwise", "orc_dequantize_rowwise",
                     "orc_double_quant", "orc_matmul_4bit", "orc_matmul_int8", "orc_linear_int8",
                     "orc_embedding_4bit", "orc_embedding_8bit", "orc_outlier_linear",
                     "orc_quantize_fp8_e4m3", "orc_dequantize_fp8_e4m3", "orc_linear_fp8"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _i64(v):
    return ctypes.c_int64(int(v))


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


def padded_cols(cols: int, blocksize: int) -> int:
    """K_padded rule of functional.py:219-221 / :260-262."""
    kp = ((cols + blocksize - 1) // blocksize) * blocksize
    if kp % 2 != 0:
        kp += blocksize
    return kp


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


# --------------------------------------------------------------------------- 4-bit
def quantize_4bit(A: torch.Tensor, blocksize: int = 64, quant_type: str = "nf4",
                  compress_statistics: bool = False, absmax: Optional[torch.Tensor] = None):
    """functional.py:163-303.  Returns (packed u8 flat, absmax, state2) where state2 is
    None or (absmax2 f32, 256) and absmax is then int8 (functional.py:288-292)."""
    A = A.contiguous()
    if A.dim() == 2:
        rows, cols = A.shape
    else:
        rows, cols = 1, A.numel()
    kp = padded_cols(cols, blocksize)
    packed = torch.zeros(rows * kp // 2, dtype=torch.uint8)
    am_out = torch.empty(rows * kp // blocksize, dtype=torch.float32)
    am_in = None if absmax is None else absmax.to(torch.float32).contiguous()
    _chk(lib().orc_quantize_4bit(_p(A), _DT[A.dtype], _i64(rows), _i64(cols), _i64(kp),
                                 int(blocksize), _QT[quant_type], _p(am_in), _p(packed), _p(am_out)),
         "quantize_4bit")
    if compress_statistics:
        q, am2 = quantize_blockwise(am_out, blocksize=256)
        return packed, q, (am2, 256)
    return packed, am_out, None


def dequantize_4bit(packed: torch.Tensor, absmax: torch.Tensor, shape, blocksize: int = 64,
                    quant_type: str = "nf4", dtype: torch.dtype = torch.float16,
                    state2=None) -> torch.Tensor:
    """functional.py:306-416 (state2 = (absmax2, blocksize2) decodes int8 absmax first, :336-337)."""
    if state2 is not None:
        absmax = dequantize_blockwise(absmax, state2[0], state2[1], torch.float32)
    shape = tuple(shape)
    if len(shape) == 2:
        rows, cols = shape
    else:
        rows, cols = 1, 1
        for s in shape:
            cols *= s
    kp = padded_cols(cols, blocksize)
    out = torch.empty(rows * cols, dtype=dtype)
    _chk(lib().orc_dequantize_4bit(_p(packed.contiguous()), _p(absmax.contiguous().float()),
                                   _i64(rows), _i64(cols), _i64(kp), int(blocksize),
                                   _QT[quant_type], _DT[dtype], _p(out)), "dequantize_4bit")
    return out.view(shape)


# --------------------------------------------------------------------------- blockwise int8
def quantize_blockwise(A: torch.Tensor, blocksize: int = 4096,
                       absmax: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """functional.py:469-539 (without `nested`).  Returns (int8 same shape, absmax f32 [nblk])."""
    A = A.contiguous()
    numel = A.numel()
    nblk = (numel + blocksize - 1) // blocksize
    out = torch.empty(A.shape, dtype=torch.int8)
    am = torch.empty(nblk, dtype=torch.float32)
    am_in = None if absmax is None else absmax.float().contiguous()
    _chk(lib().orc_quantize_blockwise(_p(A), _DT[A.dtype], _i64(numel), int(blocksize), _p(am_in),
                                      _p(out), _p(am)), "quantize_blockwise")
    return out, am


def dequantize_blockwise(q: torch.Tensor, absmax: torch.Tensor, blocksize: int = 4096,
                         dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """functional.py:542-600."""
    q = q.contiguous()
    out = torch.empty(q.shape, dtype=dtype)
    _chk(lib().orc_dequantize_blockwise(_p(q), _i64(q.numel()), _p(absmax.float().contiguous()),
                                        int(blocksize), _DT[dtype], _p(out)), "dequantize_blockwise")
    return out


# --------------------------------------------------------------------------- rowwise int8
def dequant_absmax(absmax_quant: torch.Tensor, absmax_scales: torch.Tensor, blocksize: int = 256) -> torch.Tensor:
    """functional.py:866-889, legacy form (codes [rows, num_blocks] or [num_blocks], one scale per `blocksize` codes)."""
    q = absmax_quant.contiguous()
    rows = q.shape[0] if q.dim() > 1 else 1
    num_blocks = q.numel() // rows if rows > 0 else q.numel()
    dq_blocks = absmax_scales.numel() // rows if rows > 0 else absmax_scales.numel()
    kind = 0 if q.dtype == torch.int8 else 1 if q.dtype == torch.uint8 else 2
    if kind == 2:
        q = q.float()
    out = torch.empty(q.shape, dtype=torch.float32)
    _chk(lib().orc_dequant_absmax(_p(q), kind, _i64(rows), _i64(num_blocks), _p(absmax_scales.float().contiguous()),
                                  _i64(dq_blocks), int(blocksize), _p(out)), "dequant_absmax")
    return out


def quantize_rowwise(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """functional.py:607-625."""
    t = t.contiguous()
    cols = t.shape[-1]
    rows = t.numel() // cols if cols else 0
    out = torch.empty(t.shape, dtype=torch.int8)
    scales = torch.empty(rows, dtype=torch.float32)
    _chk(lib().orc_quantize_rowwise(_p(t), _DT[t.dtype], _i64(rows), _i64(cols), _p(out), _p(scales)),
         "quantize_rowwise")
    return out, scales


def dequantize_rowwise(q: torch.Tensor, scales: torch.Tensor,
                       dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """functional.py:628-636."""
    q = q.contiguous()
    cols = q.shape[-1]
    rows = q.numel() // cols if cols else 0
    out = torch.empty(q.shape, dtype=dtype)
    _chk(lib().orc_dequantize_rowwise(_p(q), _p(scales.float().contiguous()), _i64(rows), _i64(cols),
                                      _DT[dtype], _p(out)), "dequantize_rowwise")
    return out


def double_quant(A: torch.Tensor, col_stats=None, row_stats=None):
    """functional.py:814-863.  Returns (out_col, out_row, col_stats, row_stats, None)."""
    A = A.contiguous()
    rows, cols = A.shape
    out_col = torch.empty(A.shape, dtype=torch.int8)
    out_row = torch.empty(A.shape, dtype=torch.int8)
    cs = torch.empty(cols, dtype=torch.float32) if col_stats is None else col_stats.float().contiguous().clone()
    rs = torch.empty(rows, dtype=torch.float32) if row_stats is None else row_stats.float().contiguous().clone()
    _chk(lib().orc_double_quant(_p(A), _DT[A.dtype], _i64(rows), _i64(cols), _p(out_col), _p(out_row),
                                _p(cs), _p(rs), int(col_stats is not None), int(row_stats is not None)),
         "double_quant")
    return out_col, out_row, cs, rs, None


# --------------------------------------------------------------------------- matmuls
def matmul_4bit(A: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, shape,
                blocksize: int = 64, quant_type: str = "nf4", w_dtype: torch.dtype = torch.float16,
                bias: Optional[torch.Tensor] = None, compute_dtype: Optional[torch.dtype] = None,
                state2=None) -> torch.Tensor:
    """CPU branch of functional.py:680-773 (dequantize -> F.linear in the weight dtype -> cast)."""
    if compute_dtype is None:
        compute_dtype = A.dtype
    if state2 is not None:
        absmax = dequantize_blockwise(absmax, state2[0], state2[1], torch.float32)
    N, K = shape
    kw = padded_cols(K, blocksize)
    lead = A.shape[:-1]
    A2 = A.reshape(-1, A.shape[-1]).contiguous()
    M = A2.shape[0]
    out = torch.empty(M, N, dtype=compute_dtype)
    b = None if bias is None else bias.contiguous()
    _chk(lib().orc_matmul_4bit(_p(A2), _DT[A2.dtype], _i64(M), _i64(K), _p(packed.contiguous()),
                               _p(absmax.float().contiguous()), _i64(N), _i64(kw), int(blocksize),
                               _QT[quant_type], _DT[w_dtype], _p(b), 0 if b is None else _DT[b.dtype],
                               _DT[compute_dtype], _p(out)), "matmul_4bit")
    return out.reshape(*lead, N)


def matmul_int8(A: torch.Tensor, B: torch.Tensor, A_scales: torch.Tensor, B_scales: torch.Tensor,
                dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """functional.py:788-793.  A int8 [M,K]; B int8 [K,N]; scales per row of A / column of B."""
    A = A.contiguous()
    B = B.contiguous()
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty(M, N, dtype=dtype)
    _chk(lib().orc_matmul_int8(_p(A), _p(B), _p(A_scales.float().contiguous()),
                               _p(B_scales.float().contiguous()), _i64(M), _i64(N), _i64(K), _DT[dtype],
                               _p(out)), "matmul_int8")
    return out


def linear_int8(x: torch.Tensor, weight_int8: torch.Tensor, weight_scales: torch.Tensor,
                bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Linear8bit.forward, nn/linear8bit.py:70-102 (x and bias in the compute dtype)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    M, K = x2.shape
    N = weight_int8.shape[0]
    out = torch.empty(M, N, dtype=x.dtype)
    b = None if bias is None else bias.to(x.dtype).contiguous()
    _chk(lib().orc_linear_int8(_p(x2), _DT[x.dtype], _i64(M), _i64(K), _p(weight_int8.contiguous()),
                               _p(weight_scales.float().contiguous()), _i64(N), _p(b), _p(out)),
         "linear_int8")
    return out.reshape(*lead, N)


def embedding_4bit(input: torch.Tensor, weight_packed: torch.Tensor, weight_absmax: torch.Tensor,
                   embedding_dim: int, blocksize: int = 64, quant_type: str = "nf4",
                   padding_idx: Optional[int] = None, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """Embedding4bit.forward, nn/embedding.py:83-138 (Python path)."""
    idx = input.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty(idx.numel(), embedding_dim, dtype=dtype)
    _chk(lib().orc_embedding_4bit(_p(idx), _i64(idx.numel()), _p(weight_packed.contiguous()),
                                  _p(weight_absmax.float().contiguous()), _i64(weight_packed.shape[0]),
                                  _i64(embedding_dim), int(blocksize), _QT[quant_type],
                                  int(padding_idx is not None), _i64(padding_idx if padding_idx is not None else 0),
                                  _DT[dtype], _p(out)), "embedding_4bit")
    return out.reshape(*input.shape, embedding_dim)


def embedding_8bit(input: torch.Tensor, weight_int8: torch.Tensor, weight_scales: torch.Tensor,
                   padding_idx: Optional[int] = None, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """Embedding8bit.forward, nn/embedding.py:255-268 (Python path)."""
    idx = input.reshape(-1).to(torch.int64).contiguous()
    dim = weight_int8.shape[1]
    out = torch.empty(idx.numel(), dim, dtype=dtype)
    _chk(lib().orc_embedding_8bit(_p(idx), _i64(idx.numel()), _p(weight_int8.contiguous()),
                                  _p(weight_scales.float().contiguous()), _i64(weight_int8.shape[0]), _i64(dim),
                                  int(padding_idx is not None), _i64(padding_idx if padding_idx is not None else 0),
                                  _DT[dtype], _p(out)), "embedding_8bit")
    return out.reshape(*input.shape, dim)


def outlier_linear(x: torch.Tensor, weight_int8: torch.Tensor, weight_scales: torch.Tensor,
                   outlier_indices: torch.Tensor, outlier_weights: torch.Tensor,
                   bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """OutlierAwareLinear.forward, nn/outlier_aware.py:84-146; x, outlier_weights, bias in the compute dtype."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    M, K = x2.shape
    N = weight_int8.shape[0]
    oi = outlier_indices.to(torch.int64).contiguous()
    ow = outlier_weights.to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    out = torch.empty(M, N, dtype=x.dtype)
    _chk(lib().orc_outlier_linear(_p(x2), _DT[x.dtype], _i64(M), _i64(K), _p(weight_int8.contiguous()),
                                  _p(weight_scales.float().contiguous()), _i64(N),
                                  _p(oi) if oi.numel() else None, _i64(oi.numel()),
                                  _p(ow) if oi.numel() else None, _p(b), _p(out)), "outlier_linear")
    return out.reshape(*lead, N)


def quantize_fp8_e4m3(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """functional.py:643-663 / :1086-1163 (the reference's own E4M3 encoder, restated literally)."""
    t = t.contiguous()
    rows, cols = t.shape
    out = torch.empty(rows, cols, dtype=torch.uint8)
    scales = torch.empty(rows, dtype=torch.float32)
    _chk(lib().orc_quantize_fp8_e4m3(_p(t), _DT[t.dtype], _i64(rows), _i64(cols), _p(out), _p(scales)), "quantize_fp8_e4m3")
    return out, scales


def dequantize_fp8_e4m3(q: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """functional.py:666-673 / :1166-1215."""
    q = q.contiguous()
    rows, cols = q.shape
    out = torch.empty(rows, cols, dtype=dtype)
    _chk(lib().orc_dequantize_fp8_e4m3(_p(q), _p(scales.float().contiguous()), _i64(rows), _i64(cols), _DT[dtype], _p(out)),
         "dequantize_fp8_e4m3")
    return out


def linear_fp8(x: torch.Tensor, weight_fp8: torch.Tensor, weight_scales: torch.Tensor,
               bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """matmul_fp8_e4m3 / LinearFP8.forward (functional.py:796-807); x and bias in the compute dtype."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    M, K = x2.shape
    N = weight_fp8.shape[0]
    out = torch.empty(M, N, dtype=x.dtype)
    b = None if bias is None else bias.to(x.dtype).contiguous()
    _chk(lib().orc_linear_fp8(_p(x2), _DT[x.dtype], _i64(M), _i64(K), _p(weight_fp8.contiguous()),
                              _p(weight_scales.float().contiguous()), _i64(N), _p(b), _p(out)), "linear_fp8")
    return out.reshape(*lead, N)
