"""
Functional API of the MI355X quantized-linear backend.

Mirrors the reference's ``mps_bitsandbytes/functional.py`` for the quantized-linear hot path
(same names, argument order, defaults, return types and validation messages) so that user code
and tests written against the reference read the same here.  Every operation runs as a
hand-written HIP kernel on gfx950 through the C ABI in ``include/mbnb_hip.h``; tensors must
live on a ``cuda`` (ROCm) device.  There is no CPU / pure-torch fallback: a missing native
library or a non-GPU tensor raises.

Host-side work done here is plumbing only: argument validation, shape arithmetic, output
allocation, dtype casts of activations/bias (reference: functional.py:764-766), stream lookup.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _native
from ._native import AbsmaxDesc, check, dtype_code, ptr, stream_ptr, on_device

# ============================================================================= codebooks
# reference: functional.py:21-32
NF4_CODEBOOK = torch.tensor([
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0
], dtype=torch.float32)

FP4_CODEBOOK = torch.tensor([
    0.0, 0.0625, 0.125, 0.25, 0.375, 0.5, 0.75, 1.0,
    -0.0, -0.0625, -0.125, -0.25, -0.375, -0.5, -0.75, -1.0
], dtype=torch.float32)


def create_normal_map(offset=0.9677083, use_extra_value=True):
    """NF4 codebook (bitsandbytes compatibility; reference functional.py:35-37)."""
    return NF4_CODEBOOK.clone()


def create_fp4_map(signed=True):
    """FP4 codebook (bitsandbytes compatibility; reference functional.py:40-42)."""
    return FP4_CODEBOOK.clone()


def _check_device(tensor: Tensor, operation: str = "operation"):
    """Device gate.  The reference accepts 'mps'/'cpu' (functional.py:76-83); this backend runs
    on ROCm GPUs only ('cuda' device type) and has no CPU path."""
    device_type = tensor.device.type
    if device_type != 'cuda':
        raise ValueError(
            f"mps-bitsandbytes-amd {operation} requires tensor on a 'cuda' (ROCm/HIP) device, "
            f"got '{device_type}'. Move tensor with .to('cuda')"
        )


_warned: set = set()


# matmul_4bit from 256 rows and 1.5 M outputs up: let the library dequantise the weight ONCE into a transient N x K scratch (in the weight dtype)
# and run a dense MFMA GEMM -- what the reference does for M > 512 (functional.py:753-767) -- instead of the fused kernels,
# which decode each weight tile once per 256 rows.  False keeps the fused kernels at every M (no N x K scratch) in matmul_4bit,
# linear_int8 and matmul_fp8_e4m3 alike (the FUSED_ONLY flag travels to all three entry points).  The scratch is transient:
# N x K_weight x 2 bytes per call from torch's caching allocator (32 MB for a 4096^2 layer, ~1 GB for a 128k x 4096 head).
DECODE_ONCE = True
MATMUL_FUSED_ONLY = 1   # include/mbnb_hip.h MBNB_MATMUL_FUSED_ONLY: the flags word of mbnb_matmul_4bit / mbnb_linear_int8 / mbnb_linear_fp8


def _as(t: Tensor, dtype: torch.dtype, device=None) -> Tensor:
    """`t.to(device, dtype).contiguous()` that dispatches nothing when `t` already is what is asked for"""
    if t.dtype != dtype or (device is not None and t.device != device):
        t = t.to(device=device if device is not None else t.device, dtype=dtype)
    return t if t.is_contiguous() else t.contiguous()


def _warn_once(key: str, message: str) -> None:
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def _padded(n: int, blocksize: int) -> int:
    """K_padded rule, reference functional.py:219-221 / :260-262."""
    p = ((n + blocksize - 1) // blocksize) * blocksize
    if p % 2 != 0:
        p += blocksize
    return p


# ============================================================================= QuantState
@dataclass
class QuantState:
    """
    Quantization state for dequantization (reference: functional.py:90-156; field-for-field
    identical, including ``as_dict`` keys — this is the checkpoint wire format).
    """
    absmax: Tensor
    shape: torch.Size
    code: Optional[Tensor] = None
    blocksize: int = 64
    quant_type: str = "nf4"
    dtype: torch.dtype = torch.float16
    offset: Optional[Tensor] = None
    state2: Optional['QuantState'] = None

    def __post_init__(self):
        if self.code is None:
            self.code = NF4_CODEBOOK if self.quant_type == "nf4" else FP4_CODEBOOK

    def to(self, device):
        """Move state to device (in place, recursive)."""
        self.absmax = self.absmax.to(device)
        if self.code is not None:
            self.code = self.code.to(device)
        if self.offset is not None:
            self.offset = self.offset.to(device)
        if self.state2 is not None:
            self.state2 = self.state2.to(device)
        return self

    def as_dict(self, packed=False):
        return {
            'absmax': self.absmax,
            'shape': self.shape,
            'blocksize': self.blocksize,
            'quant_type': self.quant_type,
            'dtype': self.dtype,
            'state2': self.state2.as_dict() if self.state2 else None,
        }

    @classmethod
    def from_dict(cls, state_dict, device='cpu'):
        state2 = None
        if state_dict.get('state2') is not None:
            state2 = cls.from_dict(state_dict['state2'], device)
        return cls(
            absmax=state_dict['absmax'].to(device),
            shape=state_dict['shape'],
            blocksize=state_dict.get('blocksize', 64),
            quant_type=state_dict.get('quant_type', 'nf4'),
            dtype=state_dict.get('dtype', torch.float16),
            state2=state2,
        )


def _absmax_desc(absmax: Tensor, state2: Optional[QuantState], keep: list) -> AbsmaxDesc:
    """Build the C descriptor of an absmax tensor.  One level of int8 double-quant
    (QuantState.state2 as produced by quantize_4bit, functional.py:288-292) is decoded inside the
    consuming kernel; anything else is first brought to plain f32 by dequantize_blockwise.
    `keep` collects tensors that must stay alive until the launch has been enqueued."""
    if absmax.device.type != 'cuda':
        raise ValueError(f"absmax must live on a 'cuda' (ROCm/HIP) device before a kernel launch, got '{absmax.device}'")
    if state2 is not None:
        if (absmax.dtype == torch.int8 and state2.state2 is None and state2.dtype == torch.float32
                and state2.absmax.dtype == torch.float32):
            bs2 = int(state2.blocksize)
            q = absmax if absmax.is_contiguous() else absmax.contiguous()
            # the second-level absmax follows the first level's device (a QuantState built by from_dict() defaults to
            # 'cpu'; handing the kernel a host pointer would be a GPU fault where the reference raises a device mismatch)
            a2 = state2.absmax
            if a2.device != absmax.device:
                a2 = a2.to(device=absmax.device)
            if not a2.is_contiguous():
                a2 = a2.contiguous()
            need2 = (q.numel() + bs2 - 1) // bs2 if bs2 > 0 else -1
            if bs2 <= 0 or a2.numel() < need2:
                raise ValueError(f"state2.absmax has {a2.numel()} elements, expected {need2} "
                                 f"({q.numel()} absmax codes in blocks of {bs2})")
            keep += [q, a2]
            return AbsmaxDesc(None, q.data_ptr(), a2.data_ptr(), bs2)
        state2 = _state_on(state2, absmax.device)
        absmax = dequantize_blockwise(absmax, state2)
    a = absmax
    if a.dtype != torch.float32:
        a = a.to(torch.float32)
    if not a.is_contiguous():
        a = a.contiguous()
    keep.append(a)
    return AbsmaxDesc(a.data_ptr(), None, None, 0)


def _state_on(state: QuantState, device) -> QuantState:
    """A copy of `state` (recursively) whose tensors live on `device`; the caller's object is left untouched."""
    if state.absmax.device == device and (state.state2 is None or state.state2.absmax.device == device):
        return state
    return QuantState(absmax=state.absmax.to(device), shape=state.shape, code=state.code, blocksize=state.blocksize,
                      quant_type=state.quant_type, dtype=state.dtype, offset=state.offset,
                      state2=None if state.state2 is None else _state_on(state.state2, device))


def _check_absmax_count(absmax: Tensor, rows: int, cols_padded: int, blocksize: int, what: str) -> None:
    """The kernels index absmax as [rows, cols_padded / blocksize] and receive no length: a checkpoint whose blocksize or
    shape disagrees with its absmax must fail here (the reference's absmax.view(N, num_blocks_per_row) raises,
    functional.py:371-373), not read past the allocation."""
    need = rows * (cols_padded // blocksize)
    if absmax.numel() != need:
        raise ValueError(f"{what}: absmax has {absmax.numel()} elements, expected {need} "
                         f"({rows} rows x {cols_padded // blocksize} blocks of {blocksize})")


# ============================================================================= 4-bit
def quantize_4bit(
    A: Tensor,
    absmax: Optional[Tensor] = None,
    out: Optional[Tensor] = None,
    blocksize: int = 64,
    compress_statistics: bool = False,
    quant_type: str = "nf4",
    quant_storage: torch.dtype = torch.uint8,
) -> Tuple[Tensor, QuantState]:
    """
    Quantize tensor to 4-bit NF4 or FP4 (reference: functional.py:163-303, bit-exact).

    2-D tensors [N, K] are quantized row-wise (each row padded to a multiple of `blocksize`);
    other shapes over the flattened tensor.  Returns (packed uint8 flat tensor, QuantState).
    """
    if quant_type not in ("nf4", "fp4"):
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
    _check_device(A, "quantize_4bit")
    if blocksize <= 0:
        raise ValueError(f"blocksize must be positive, got {blocksize}")
    if blocksize > 65536:
        raise ValueError(f"blocksize too large ({blocksize}), max is 65536")
    if (blocksize & (blocksize - 1)) != 0:
        raise ValueError(f"blocksize must be a power of 2, got {blocksize}")
    max_safe_numel = 2**31 - 1
    if A.numel() > max_safe_numel:
        raise ValueError(f"Tensor too large ({A.numel()} elements), max is {max_safe_numel}")

    orig_shape = A.shape
    orig_dtype = A.dtype
    code = dtype_code(orig_dtype, "quantize_4bit")
    A = A.contiguous()
    if A.dim() == 2:
        rows, cols = A.shape
    else:
        rows, cols = 1, A.numel()
    cols_padded = _padded(cols, blocksize)
    nbytes = rows * cols_padded // 2
    nblocks = rows * cols_padded // blocksize

    user_out = out
    if (out is not None and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == nbytes
            and out.device == A.device):
        packed = out.view(-1)
    else:
        packed = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    if blocksize == 1:
        packed.zero_()
    absmax_in = None
    if absmax is not None:
        absmax_in = absmax.to(device=A.device, dtype=torch.float32).contiguous().view(-1)
        if absmax_in.numel() != nblocks:
            raise ValueError(f"absmax has {absmax_in.numel()} elements, expected {nblocks}")
    # compress_statistics: the 256-block int8 quantisation of absmax (functional.py:288-292) is fused into the same launch
    # when the library supports the blocksize; the f32 absmax then never exists in memory
    fused_dq = compress_statistics and absmax_in is None and 8 <= blocksize <= 512 and nblocks > 0
    absmax_t = None
    state2 = None
    with on_device(A.device):
        if fused_dq:
            codes = torch.empty(nblocks, dtype=torch.int8, device=A.device)
            absmax2 = torch.empty((nblocks + 255) // 256, dtype=torch.float32, device=A.device)
            check(_native.lib().mbnb_quantize_4bit_dq(
                ptr(A), code, rows, cols, cols_padded, int(blocksize), _native.QUANT_CODE[quant_type], ptr(packed),
                ptr(codes), ptr(absmax2), stream_ptr(A.device)), "quantize_4bit")
            absmax_t = codes
            state2 = QuantState(absmax=absmax2, shape=torch.Size([nblocks]), blocksize=256, quant_type="int8",
                                dtype=torch.float32)     # = what quantize_blockwise(absmax, blocksize=256) returns
        else:
            absmax_out = torch.empty(nblocks, dtype=torch.float32, device=A.device)
            check(_native.lib().mbnb_quantize_4bit(
                ptr(A), code, rows, cols, cols_padded, int(blocksize), _native.QUANT_CODE[quant_type],
                ptr(absmax_in), ptr(packed), ptr(absmax_out), stream_ptr(A.device)), "quantize_4bit")
            absmax_t = absmax_out

    if user_out is not None and packed.data_ptr() != user_out.data_ptr():
        user_out.view(-1)[:] = packed.to(user_out.dtype)  # reference: out[:] = ... (functional.py:251)
        packed = user_out.flatten()
    elif quant_storage != torch.uint8:
        packed = packed.to(quant_storage)

    if compress_statistics and not fused_dq:
        absmax_t, state2 = quantize_blockwise(absmax_t, blocksize=256)

    quant_state = QuantState(
        absmax=absmax_t,
        shape=orig_shape,
        blocksize=blocksize,
        quant_type=quant_type,
        dtype=orig_dtype,
        state2=state2,
    )
    return packed, quant_state


def dequantize_4bit(
    A: Tensor,
    quant_state: Optional[QuantState] = None,
    absmax: Optional[Tensor] = None,
    out: Optional[Tensor] = None,
    blocksize: int = 64,
    quant_type: str = "nf4",
) -> Tensor:
    """Dequantize a 4-bit tensor (reference: functional.py:306-416, bit-exact)."""
    _check_device(A, "dequantize_4bit")
    state2 = None
    if quant_state is not None:
        absmax = quant_state.absmax
        blocksize = quant_state.blocksize
        quant_type = quant_state.quant_type
        shape = quant_state.shape
        dtype = quant_state.dtype
        state2 = quant_state.state2
    else:
        if absmax is None:
            raise ValueError("Either quant_state or absmax must be provided")
        shape = None
        dtype = torch.float16
    if quant_type not in ("nf4", "fp4"):
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")

    A = A.contiguous()
    if A.dtype != torch.uint8:
        A = A.to(torch.uint8)
    if shape is not None and len(shape) == 2:
        rows, cols = int(shape[0]), int(shape[1])
        cols_padded = _padded(cols, blocksize)
        out_shape = (rows, cols)
    else:
        # flat layout: the number of blocks is what absmax says (functional.py:399-401)
        rows = 1
        cols_padded = absmax.numel() * blocksize
        if shape is not None:
            cols = int(torch.Size(shape).numel())
            out_shape = tuple(shape)
        elif out is not None:
            cols = out.numel()
            out_shape = tuple(out.shape)
        else:
            cols = cols_padded
            out_shape = (cols,)
    if A.numel() * 2 < rows * cols_padded or cols > cols_padded:
        raise ValueError(
            f"packed tensor has {A.numel()} bytes but absmax/shape describe {rows * cols_padded} 4-bit values")
    _check_absmax_count(absmax, rows, cols_padded, blocksize, "dequantize_4bit")

    keep: list = []
    desc = _absmax_desc(absmax.to(A.device), state2, keep)
    direct = (out is not None and out.dtype == dtype and out.is_contiguous() and out.numel() == rows * cols
              and out.device == A.device)
    result = out if direct else torch.empty(out_shape, dtype=dtype, device=A.device)
    with on_device(A.device):
        check(_native.lib().mbnb_dequantize_4bit(
            ptr(A), ctypes.byref(desc), rows, cols, cols_padded, int(blocksize), _native.QUANT_CODE[quant_type],
            dtype_code(dtype, "dequantize_4bit"), ptr(result), stream_ptr(A.device)), "dequantize_4bit")
    if out is not None and not direct:
        out[:] = result.view(out.shape).to(out.dtype)
        return out
    return result


def quantize_nf4(A: Tensor, absmax: Optional[Tensor] = None, out: Optional[Tensor] = None, blocksize: int = 64,
                 compress_statistics: bool = False, quant_storage: torch.dtype = torch.uint8):
    """Alias for quantize_4bit with quant_type='nf4' (reference functional.py:419-428)."""
    return quantize_4bit(A, absmax, out, blocksize, compress_statistics, "nf4", quant_storage)


def dequantize_nf4(A: Tensor, quant_state: Optional[QuantState] = None, absmax: Optional[Tensor] = None,
                   out: Optional[Tensor] = None, blocksize: int = 64) -> Tensor:
    """Alias for dequantize_4bit with quant_type='nf4' (reference functional.py:431-439)."""
    return dequantize_4bit(A, quant_state, absmax, out, blocksize, "nf4")


def quantize_fp4(A: Tensor, absmax: Optional[Tensor] = None, out: Optional[Tensor] = None, blocksize: int = 64,
                 compress_statistics: bool = False, quant_storage: torch.dtype = torch.uint8):
    """Alias for quantize_4bit with quant_type='fp4' (reference functional.py:442-451)."""
    return quantize_4bit(A, absmax, out, blocksize, compress_statistics, "fp4", quant_storage)


def dequantize_fp4(A: Tensor, quant_state: Optional[QuantState] = None, absmax: Optional[Tensor] = None,
                   out: Optional[Tensor] = None, blocksize: int = 64) -> Tensor:
    """Alias for dequantize_4bit with quant_type='fp4' (reference functional.py:454-462)."""
    return dequantize_4bit(A, quant_state, absmax, out, blocksize, "fp4")


# ============================================================================= blockwise int8
def quantize_blockwise(
    A: Tensor,
    code: Optional[Tensor] = None,
    absmax: Optional[Tensor] = None,
    out: Optional[Tensor] = None,
    blocksize: int = 4096,
    nested: bool = False,
) -> Tuple[Tensor, QuantState]:
    """INT8 blockwise absmax quantization (reference: functional.py:469-539, bit-exact)."""
    _check_device(A, "quantize_blockwise")
    if blocksize <= 0:
        raise ValueError(f"blocksize must be positive, got {blocksize}")
    if blocksize > 65536:
        raise ValueError(f"blocksize too large ({blocksize}), max is 65536")
    orig_shape = A.shape
    orig_dtype = A.dtype
    dcode = dtype_code(orig_dtype, "quantize_blockwise")
    A = A.contiguous()
    numel = A.numel()
    nblocks = (numel + blocksize - 1) // blocksize
    absmax_in = None
    if absmax is not None:
        absmax_in = absmax.to(device=A.device, dtype=torch.float32).contiguous().view(-1)
    q = torch.empty(numel, dtype=torch.int8, device=A.device)
    absmax_out = torch.empty(nblocks, dtype=torch.float32, device=A.device)
    with on_device(A.device):
        check(_native.lib().mbnb_quantize_blockwise(
            ptr(A), dcode, numel, int(blocksize), ptr(absmax_in), ptr(q), ptr(absmax_out),
            stream_ptr(A.device)), "quantize_blockwise")
    if out is not None:
        out.view(-1)[:numel] = q  # reference writes the blocked result into `out` (functional.py:522)
    q = q.view(orig_shape)

    absmax_t = absmax_out
    state2 = None
    if nested:
        absmax_t, state2 = quantize_blockwise(absmax_out, blocksize=256)
    quant_state = QuantState(
        absmax=absmax_t,
        shape=orig_shape,
        blocksize=blocksize,
        quant_type="int8",
        dtype=orig_dtype,
        state2=state2,
    )
    return q, quant_state


def dequantize_blockwise(
    A: Tensor,
    quant_state: Optional[QuantState] = None,
    absmax: Optional[Tensor] = None,
    code: Optional[Tensor] = None,
    out: Optional[Tensor] = None,
    blocksize: int = 4096,
    nested: bool = False,
) -> Tensor:
    """Dequantize blockwise INT8 (reference: functional.py:542-600, bit-exact)."""
    _check_device(A, "dequantize_blockwise")
    if quant_state is not None:
        absmax = quant_state.absmax
        blocksize = quant_state.blocksize
        shape = quant_state.shape
        dtype = quant_state.dtype
        if quant_state.state2 is not None:
            absmax = dequantize_blockwise(absmax, quant_state.state2)
    else:
        if absmax is None:
            raise ValueError("Either quant_state or absmax must be provided")
        shape = A.shape
        dtype = torch.float16
    q = A.contiguous()
    if q.dtype != torch.int8:
        q = q.to(torch.int8)
    numel = q.numel()
    am = absmax.to(device=A.device, dtype=torch.float32).contiguous()
    result = torch.empty(numel, dtype=dtype, device=A.device)
    with on_device(A.device):
        check(_native.lib().mbnb_dequantize_blockwise(
            ptr(q), numel, ptr(am), int(blocksize), dtype_code(dtype, "dequantize_blockwise"), ptr(result),
            stream_ptr(A.device)), "dequantize_blockwise")
    result = result.view(shape)
    if out is not None:
        out[:] = result
        return out
    return result


# ============================================================================= rowwise int8
def quantize_rowwise(tensor: Tensor) -> Tuple[Tensor, Tensor]:
    """Row-wise absmax INT8 quantization (reference: functional.py:607-625, bit-exact).
    Returns (int8 tensor of the same shape, scales f32 [rows]); scales hold the absmax itself."""
    _check_device(tensor, "quantize_rowwise")
    orig_shape = tensor.shape
    t = tensor.contiguous()
    cols = t.shape[-1]
    rows = t.numel() // cols if cols else 0
    q = torch.empty(t.shape, dtype=torch.int8, device=t.device)
    scales = torch.empty(rows, dtype=torch.float32, device=t.device)
    with on_device(t.device):
        check(_native.lib().mbnb_quantize_rowwise(
            ptr(t), dtype_code(t.dtype, "quantize_rowwise"), rows, cols, ptr(q), ptr(scales),
            stream_ptr(t.device)), "quantize_rowwise")
    return q.view(orig_shape), scales


def dequantize_rowwise(quantized: Tensor, scales: Tensor, dtype: torch.dtype = torch.float16) -> Tensor:
    """Dequantize row-wise INT8 (reference: functional.py:628-636, bit-exact)."""
    _check_device(quantized, "dequantize_rowwise")
    orig_shape = quantized.shape
    q = quantized.contiguous()
    cols = q.shape[-1]
    rows = q.numel() // cols if cols else 0
    s = scales.to(device=q.device, dtype=torch.float32).contiguous().view(-1)
    if s.numel() != rows:
        raise ValueError(f"scales has {s.numel()} elements, expected {rows}")
    out = torch.empty(q.shape, dtype=dtype, device=q.device)
    with on_device(q.device):
        check(_native.lib().mbnb_dequantize_rowwise(
            ptr(q), ptr(s), rows, cols, dtype_code(dtype, "dequantize_rowwise"), ptr(out),
            stream_ptr(q.device)), "dequantize_rowwise")
    return out.view(orig_shape)


# ============================================================================= matmuls
def matmul_4bit(
    A: Tensor,
    B: Tensor,
    quant_state: QuantState,
    bias: Optional[Tensor] = None,
    compute_dtype: Optional[torch.dtype] = None,
) -> Tensor:
    """
    Matrix multiplication with 4-bit quantized weights: ``A[..., K] @ dequant(B)[N, K]^T + bias``.

    Signature of the reference (functional.py:680-773).  Below 256 rows (1.5 M outputs) ONE fused HIP kernel decodes the
    packed weight inside the GEMV / MFMA GEMM (the reference fuses only on MPS and only for M <= 512); from there up the
    library does the reference's own two steps (functional.py:753-767) on a transient scratch: ``dequantize_4bit`` once, then
    a dense MFMA GEMM (``DECODE_ONCE = False`` keeps the fused kernels at every M).  Either way the decoded weight is
    rounded to ``quant_state.dtype`` exactly as ``dequantize_4bit`` would, the contraction runs in that dtype with f32
    accumulation, and the result is cast to ``compute_dtype`` (default ``A.dtype``) -- the numerics of the reference's CPU
    branch (functional.py:756-773).
    """
    if compute_dtype is None:
        compute_dtype = A.dtype
    _check_device(A, "matmul_4bit")
    _check_device(B, "matmul_4bit")
    if len(quant_state.shape) != 2:
        raise ValueError(f"matmul_4bit needs a 2-D quantized weight, got shape {tuple(quant_state.shape)}")
    N, K = int(quant_state.shape[0]), int(quant_state.shape[1])
    if A.shape[-1] != K:
        raise RuntimeError(
            f"mat1 and mat2 shapes cannot be multiplied ({A.numel() // max(A.shape[-1], 1)}x{A.shape[-1]} and {K}x{N})")
    blocksize = quant_state.blocksize
    K_weight = _padded(K, blocksize)
    w_dtype = quant_state.dtype
    w_code = dtype_code(w_dtype, "matmul_4bit")

    orig_shape = A.shape
    # reference: A.to(weight.dtype), functional.py:764.  (No-op conversions are skipped by hand: at M = 1 the kernel takes 5 us and every
    # dispatched torch call costs ~1 us of host time, tools/host_overhead.py.)
    A2 = A if A.dim() == 2 else A.reshape(-1, K)
    if A2.dtype != w_dtype:
        A2 = A2.to(w_dtype)
    if not A2.is_contiguous():
        A2 = A2.contiguous()
    M = A2.shape[0]
    bias_w = None
    if bias is not None:
        bias_w = bias   # functional.py:765-766
        if bias_w.dtype != w_dtype or bias_w.device != A.device:
            bias_w = bias_w.to(device=A.device, dtype=w_dtype)
        if not bias_w.is_contiguous():
            bias_w = bias_w.contiguous()
    packed = B if B.is_contiguous() else B.contiguous()
    if packed.dtype != torch.uint8:
        packed = packed.to(torch.uint8)
    if packed.numel() * 2 < N * K_weight:
        raise ValueError(f"packed weight has {packed.numel()} bytes, expected {N * K_weight // 2}")
    _check_absmax_count(quant_state.absmax, N, K_weight, blocksize, "matmul_4bit")

    if M * N * K >= (1 << 27) and ((w_dtype == torch.float32 and (K % 4 != 0 or not DECODE_ONCE)) or
                                   (w_dtype != torch.float32 and (blocksize < 32 or K % 8 != 0))):
        # the 16-bit MFMA / weight-streaming kernels need blocksize >= 32 and K % 8 == 0, the f32 decode-once path K % 4 == 0;
        # anything else runs on the generic one-wave-per-output-column kernel, orders of magnitude slower at this size
        _warn_once("matmul_4bit.generic",
                   f"matmul_4bit: weight dtype {w_dtype}, blocksize {blocksize}, K {K} takes the generic (non-MFMA) kernel; "
                   f"quantize a float16 / bfloat16 weight with blocksize >= 32 and K % 8 == 0 for the fast paths "
                   f"(e.g. Linear4bit.from_linear(linear.to(torch.bfloat16)))")
    out_dtype = compute_dtype if compute_dtype in _native.DTYPE_CODE else w_dtype
    out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    keep: list = []
    am = quant_state.absmax
    desc = _absmax_desc(am if am.device == A.device else am.to(A.device), quant_state.state2, keep)
    # The library's scratch (0 bytes = none needed for this shape): mid-sized M leaves too few output tiles for 256 CUs and
    # K is split over f32 partials; large M (>= 256 rows, >= 1.5 M outputs) decodes the weight ONCE into it (N x K_weight in the weight dtype) and
    # runs a dense MFMA GEMM instead of re-decoding every weight tile per 256 rows.  torch's caching allocator makes the
    # allocation a pointer bump; the memory goes back to the pool on return.
    flags = 0 if DECODE_ONCE else MATMUL_FUSED_ONLY
    ws_bytes = _matmul4_ws_bytes(M, N, K, K_weight, w_code, flags)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes > 0 else None
    with on_device(A.device):
        check(_native.lib().mbnb_matmul_4bit(
            ptr(A2), M, K, ptr(packed), ctypes.byref(desc), N, K_weight, int(blocksize),
            _native.QUANT_CODE[quant_state.quant_type], w_code, ptr(bias_w), _native.DTYPE_CODE[out_dtype],
            ptr(out), ptr(ws), ws_bytes, flags, stream_ptr(A.device)), "matmul_4bit")
    if out_dtype != compute_dtype:
        out = out.to(compute_dtype)
    return out if len(orig_shape) == 2 else out.reshape(*orig_shape[:-1], N)


_WS_CACHE: dict = {}


def _matmul4_ws_bytes(M: int, N: int, K: int, K_weight: int, w_code: int, flags: int) -> int:
    """mbnb_matmul_4bit_workspace_bytes, memoised: a pure function of its arguments (a decode loop asks the same question for every token)."""
    key = (M, N, K, K_weight, w_code, flags)
    v = _WS_CACHE.get(key)
    if v is None:
        if len(_WS_CACHE) > 4096:
            _WS_CACHE.clear()
        v = _WS_CACHE[key] = int(_native.lib().mbnb_matmul_4bit_workspace_bytes(M, N, K, K_weight, w_code, flags))
    return v


def matmul_nf4(input: Tensor, weight_packed: Tensor, weight_state: QuantState,
               bias: Optional[Tensor] = None) -> Tensor:
    """Matrix multiplication with NF4-quantized weights (reference functional.py:776-779)."""
    return matmul_4bit(input, weight_packed, weight_state, bias)


def matmul_fp4(input: Tensor, weight_packed: Tensor, weight_state: QuantState,
               bias: Optional[Tensor] = None) -> Tensor:
    """Matrix multiplication with FP4-quantized weights (reference functional.py:782-785)."""
    return matmul_4bit(input, weight_packed, weight_state, bias)


def matmul_int8(A: Tensor, B: Tensor, A_scales: Tensor, B_scales: Tensor,
                dtype: torch.dtype = torch.float16) -> Tensor:
    """
    INT8 matmul with fused dequantization (reference: functional.py:788-793).

    A int8 [M, K] with one scale per row; B int8 [K, N] with one scale per column.  Computed as
    ``int32(A·B) * (sA[m]/127) * (sB[n]/127)`` on the int8 MFMA (what the reference's Metal kernel
    mm:155-196 computes; equal to the reference CPU formulation within 4e-4 rel-err).
    """
    _check_device(A, "matmul_int8")
    _check_device(B, "matmul_int8")
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[0]:
        raise RuntimeError(f"matmul_int8: shapes {tuple(A.shape)} and {tuple(B.shape)} cannot be multiplied")
    A = A.contiguous()
    B = B.contiguous()
    if A.dtype != torch.int8 or B.dtype != torch.int8:
        raise TypeError("matmul_int8: A and B must be int8")
    M, K = A.shape
    N = B.shape[1]
    sa = A_scales.to(device=A.device, dtype=torch.float32).contiguous().view(-1)
    sb = B_scales.to(device=A.device, dtype=torch.float32).contiguous().view(-1)
    if sa.numel() != M or sb.numel() != N:
        raise ValueError(f"matmul_int8: need {M} A_scales and {N} B_scales, got {sa.numel()} and {sb.numel()}")
    out_dtype = dtype if dtype in _native.DTYPE_CODE else torch.float32
    out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    # large aligned problems are read in place ([K, N] through a transposing LDS read); the rest need an N*K-byte scratch
    ws_bytes = int(_native.lib().mbnb_matmul_int8_workspace_bytes(M, N, K))
    if not DECODE_ONCE and K % 128 == 0 and N % 16 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 96:
        ws_bytes = 0      # large aligned problems without scratch: B read in place as [K, N] (no transpose)
    workspace = torch.empty(ws_bytes, dtype=torch.int8, device=A.device) if ws_bytes > 0 else None
    with on_device(A.device):
        check(_native.lib().mbnb_matmul_int8(
            ptr(A), ptr(B), ptr(sa), ptr(sb), M, N, K, _native.DTYPE_CODE[out_dtype], ptr(out), ptr(workspace), ws_bytes,
            stream_ptr(A.device)), "matmul_int8")
    return out if out_dtype == dtype else out.to(dtype)


def linear_int8(input: Tensor, weight_int8: Tensor, weight_scales: Tensor, bias: Optional[Tensor] = None,
                dtype: Optional[torch.dtype] = None) -> Tensor:
    """
    ``input[..., K] @ dequantize_rowwise(weight_int8[N, K], weight_scales, dtype)^T + bias`` in one
    kernel — the forward of Linear8bit (reference: nn/linear8bit.py:70-102; native binding
    `_C.linear_int8`, mm:1836-1886).
    """
    _check_device(input, "linear_int8")
    _check_device(weight_int8, "linear_int8")
    if dtype is None:
        dtype = input.dtype
    dcode = dtype_code(dtype, "linear_int8")
    N, K = weight_int8.shape
    if input.shape[-1] != K:
        raise RuntimeError(f"linear_int8: input width {input.shape[-1]} does not match weight {tuple(weight_int8.shape)}")
    x = _as(input if input.dim() == 2 else input.reshape(-1, K), dtype)      # (no-op conversions skipped: ~1 us of host time each)
    M = x.shape[0]
    w = weight_int8 if weight_int8.is_contiguous() else weight_int8.contiguous()
    s = _as(weight_scales, torch.float32, x.device)
    b = None if bias is None else _as(bias, dtype, x.device)
    out = torch.empty(M, N, dtype=dtype, device=x.device)
    flags = 0 if DECODE_ONCE else MATMUL_FUSED_ONLY
    ws_bytes = int(_native.lib().mbnb_linear_int8_workspace_bytes(M, N, K, flags)) if M > 16 else 0   # split-K partials / the dequantised weight
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
    with on_device(x.device):
        check(_native.lib().mbnb_linear_int8(ptr(x), dcode, M, K, ptr(w), ptr(s), N, ptr(b), ptr(out), ptr(ws), ws_bytes,
                                             flags, stream_ptr(x.device)), "linear_int8")
    return out if input.dim() == 2 else out.reshape(*input.shape[:-1], N)


def linear_dense(input: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """
    ``input[..., K] @ weight[N, K]^T + bias`` on an ALREADY dequantised 16-bit weight: the dense MFMA GEMM the decode-once
    paths run after their dequantise pass (csrc/gemm_dense.h), for callers that keep the dequantised weight -- Linear8bit's
    cache (reference nn/linear8bit.py:70-102: `F.linear(x, self._get_weight(), bias)`).  Same slice plan, hence the same bits,
    as ``linear_int8`` / ``matmul_4bit`` on the quantised weight at that shape.  K % 64 == 0, K >= 128.
    """
    _check_device(input, "linear_dense")
    _check_device(weight, "linear_dense")
    dtype = weight.dtype
    dcode = dtype_code(dtype, "linear_dense")
    N, K = weight.shape
    if input.shape[-1] != K:
        raise RuntimeError(f"linear_dense: input width {input.shape[-1]} does not match weight {tuple(weight.shape)}")
    x = input.reshape(-1, K).to(dtype).contiguous()
    M = x.shape[0]
    w = weight.contiguous()
    b = None if bias is None else bias.to(device=x.device, dtype=dtype).contiguous()
    out = torch.empty(M, N, dtype=dtype, device=x.device)
    ws_bytes = int(_native.lib().mbnb_gemm_dense_workspace_bytes(M, N, K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
    with on_device(x.device):
        check(_native.lib().mbnb_gemm_dense(ptr(x), ptr(w), dcode, ptr(b), dcode, ptr(out), M, N, K, K, ptr(ws), ws_bytes, 0,
                                            stream_ptr(x.device)), "linear_dense")
    return out.reshape(*input.shape[:-1], N)


def dense_path_applies(M: int, N: int, K: int) -> bool:
    """True where ``linear_int8`` / ``matmul_4bit`` take the decode-once path (dequantise into scratch + dense GEMM)."""
    return bool(_native.lib().mbnb_gemm_dense_applies(M, N, K, K))


# ============================================================================= FP8 E4M3 (the reference's own format)
def quantize_fp8_e4m3(tensor: Tensor) -> Tuple[Tensor, Tensor]:
    """
    Row-wise FP8 E4M3 quantization in the reference's format (functional.py:643-663, encoder :1086-1163), bit-exact:
    returns (uint8 [rows, cols], scales f32 [rows]) with scales = clamp(max|row| / 448, 1e-12).  The encoder is the
    reference's, not the OCP conversion (no mantissa carry, subnormals flushed, |v| >= 256 saturates to 240).
    """
    if tensor.dim() != 2:
        raise ValueError("Input must be 2D")
    _check_device(tensor, "quantize_fp8_e4m3")
    A = tensor if tensor.dtype in _native.DTYPE_CODE else tensor.float()
    A = A.contiguous()
    rows, cols = A.shape
    out = torch.empty(rows, cols, dtype=torch.uint8, device=A.device)
    scales = torch.empty(rows, dtype=torch.float32, device=A.device)
    with on_device(A.device):
        check(_native.lib().mbnb_quantize_fp8_e4m3(ptr(A), dtype_code(A.dtype, "quantize_fp8_e4m3"), rows, cols, ptr(out),
                                                   ptr(scales), stream_ptr(A.device)), "quantize_fp8_e4m3")
    return out, scales


def dequantize_fp8_e4m3(quantized: Tensor, scales: Tensor, dtype: torch.dtype = torch.float16) -> Tensor:
    """decode(byte) * scales[row] -> dtype (functional.py:666-673, decoder :1166-1215), bit-exact."""
    _check_device(quantized, "dequantize_fp8_e4m3")
    q = quantized.contiguous()
    rows, cols = q.shape
    s = scales.to(device=q.device, dtype=torch.float32).contiguous()
    out = torch.empty(rows, cols, dtype=dtype, device=q.device)
    with on_device(q.device):
        check(_native.lib().mbnb_dequantize_fp8_e4m3(ptr(q), ptr(s), rows, cols, dtype_code(dtype, "dequantize_fp8_e4m3"),
                                                     ptr(out), stream_ptr(q.device)), "dequantize_fp8_e4m3")
    return out


def matmul_fp8_e4m3(input: Tensor, weight: Tensor, weight_scales: Tensor, bias: Optional[Tensor] = None,
                    dtype: torch.dtype = torch.float16) -> Tensor:
    """
    ``input[..., K] @ dequantize_fp8_e4m3(weight[N, K], weight_scales, dtype)^T + bias`` (functional.py:796-807) in one
    kernel: the W8A16 GEMMs of linear_int8 with the FP8 byte decoder in the weight producer.
    """
    _check_device(input, "matmul_fp8_e4m3")
    _check_device(weight, "matmul_fp8_e4m3")
    dcode = dtype_code(dtype, "matmul_fp8_e4m3")
    N, K = weight.shape
    is_1d = input.dim() == 1
    x = (input.unsqueeze(0) if is_1d else input)
    if x.shape[-1] != K:
        raise RuntimeError(f"matmul_fp8_e4m3: input width {x.shape[-1]} does not match weight {tuple(weight.shape)}")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K).to(dtype).contiguous()
    M = x2.shape[0]
    w = weight.contiguous()
    s = weight_scales.to(device=x2.device, dtype=torch.float32).contiguous()
    b = None if bias is None else bias.to(device=x2.device, dtype=dtype).contiguous()
    out = torch.empty(M, N, dtype=dtype, device=x2.device)
    flags = 0 if DECODE_ONCE else MATMUL_FUSED_ONLY
    ws_bytes = int(_native.lib().mbnb_linear_int8_workspace_bytes(M, N, K, flags)) if M > 16 else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x2.device) if ws_bytes > 0 else None
    with on_device(x2.device):
        check(_native.lib().mbnb_linear_fp8(ptr(x2), dcode, M, K, ptr(w), ptr(s), N, ptr(b), ptr(out), ptr(ws), ws_bytes,
                                            flags, stream_ptr(x2.device)), "matmul_fp8_e4m3")
    out = out.reshape(*lead, N)
    return out.squeeze(0) if is_1d else out


# ============================================================================= quantized embedding lookups
def embedding_4bit(input: Tensor, weight_packed: Tensor, weight_absmax: Tensor, embedding_dim: int,
                   blocksize: int = 64, quant_type: str = 'nf4', padding_idx: Optional[int] = None,
                   dtype: torch.dtype = torch.float16) -> Tensor:
    """
    Gather + dequantize rows of a 4-bit embedding table in one kernel: the forward of Embedding4bit
    (reference: nn/embedding.py:83-138; native bindings `_C.embedding_4bit_nf4/_fp4`, mm:2309-2388).
    `weight_packed` u8 [num, dim/2], `weight_absmax` f32 [num, ceil(dim/blocksize)]; returns
    ``input.shape + [embedding_dim]`` in `dtype`, rows equal to `padding_idx` zeroed.  Bit-exact against the
    reference's Python path.  Indices are not range-checked on the host (no device synchronisation);
    out-of-range rows come back as zeros.
    """
    _check_device(weight_packed, "embedding_4bit")
    if quant_type not in _native.QUANT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
    dev = weight_packed.device
    idx = input.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    num = weight_packed.shape[0]
    out = torch.empty(idx.numel(), embedding_dim, dtype=dtype, device=dev)
    wp = weight_packed.contiguous()
    wa = weight_absmax.to(device=dev, dtype=torch.float32).contiguous()
    with on_device(dev):
        check(_native.lib().mbnb_embedding_4bit(ptr(idx), idx.numel(), ptr(wp), ptr(wa), num, int(embedding_dim),
                                                int(blocksize), _native.QUANT_CODE[quant_type], int(padding_idx is not None),
                                                int(padding_idx) if padding_idx is not None else 0,
                                                dtype_code(dtype, "embedding_4bit"), ptr(out), stream_ptr(dev)),
              "embedding_4bit")
    return out.reshape(*input.shape, embedding_dim)


def embedding_8bit(input: Tensor, weight_int8: Tensor, weight_scales: Tensor, padding_idx: Optional[int] = None,
                   dtype: torch.dtype = torch.float16) -> Tensor:
    """
    Gather + dequantize rows of an int8 embedding table: the forward of Embedding8bit
    (reference: nn/embedding.py:255-268; native binding `_C.embedding_8bit`, mm:2390-2427).  Bit-exact against the
    reference's Python path: ``q.to(dtype) * (scale / 127.0).to(dtype)``.
    """
    _check_device(weight_int8, "embedding_8bit")
    dev = weight_int8.device
    idx = input.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    num, dim = weight_int8.shape
    out = torch.empty(idx.numel(), dim, dtype=dtype, device=dev)
    w = weight_int8.contiguous()
    s = weight_scales.to(device=dev, dtype=torch.float32).contiguous()
    with on_device(dev):
        check(_native.lib().mbnb_embedding_8bit(ptr(idx), idx.numel(), ptr(w), ptr(s), num, dim,
                                                int(padding_idx is not None),
                                                int(padding_idx) if padding_idx is not None else 0,
                                                dtype_code(dtype, "embedding_8bit"), ptr(out), stream_ptr(dev)),
              "embedding_8bit")
    return out.reshape(*input.shape, dim)


# ============================================================================= outlier-aware INT8 linear
def outlier_linear(input: Tensor, weight_int8: Tensor, weight_scales: Tensor, outlier_indices: Tensor,
                   outlier_weights: Tensor, bias: Optional[Tensor] = None,
                   dtype: Optional[torch.dtype] = None) -> Tensor:
    """
    The forward of OutlierAwareLinear (reference: nn/outlier_aware.py:84-146): row-wise INT8 quantisation of the
    non-outlier input columns, int8 x int8 contraction on the MFMA, the outlier columns in `dtype`, bias.
    The reference multiplies dtype-rounded dequantised operands; the exact integer contraction used here differs
    from it by <= 4e-4 (fp16) / 1.3e-3 (bf16) relative (Frobenius).
    """
    _check_device(input, "outlier_linear")
    _check_device(weight_int8, "outlier_linear")
    if dtype is None:
        dtype = input.dtype
    dcode = dtype_code(dtype, "outlier_linear")
    N, K = weight_int8.shape
    if input.shape[-1] != K:
        raise RuntimeError(f"outlier_linear: input width {input.shape[-1]} does not match weight {tuple(weight_int8.shape)}")
    dev = input.device
    x = input.reshape(-1, K).to(dtype).contiguous()
    M = x.shape[0]
    w = weight_int8.contiguous()
    s = weight_scales.to(device=dev, dtype=torch.float32).contiguous()
    n_out = int(outlier_indices.numel())
    oi = outlier_indices.to(device=dev, dtype=torch.int64).contiguous() if n_out else None
    ow = outlier_weights.to(device=dev, dtype=dtype).contiguous() if n_out else None
    b = None if bias is None else bias.to(device=dev, dtype=dtype).contiguous()
    out = torch.empty(M, N, dtype=dtype, device=dev)
    ws_bytes = int(_native.lib().mbnb_outlier_linear_workspace_bytes(M, K, n_out))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(_native.lib().mbnb_outlier_linear(ptr(x), dcode, M, K, ptr(w), ptr(s), N, ptr(oi), n_out, ptr(ow), ptr(b),
                                                   ptr(out), ptr(ws), ws_bytes, stream_ptr(dev)), "outlier_linear")
    return out.reshape(*input.shape[:-1], N)


# ============================================================================= double quant (LLM.int8 stats)
def double_quant(
    A: Tensor,
    col_stats: Optional[Tensor] = None,
    row_stats: Optional[Tensor] = None,
    out_col: Optional[Tensor] = None,
    out_row: Optional[Tensor] = None,
    threshold: float = 0.0,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """
    Row- and column-wise INT8 quantization with their absmax statistics
    (reference: functional.py:814-863, bit-exact).  Returns
    (col_quantized, row_quantized, col_stats, row_stats, None); `threshold` is ignored as in the
    reference, and caller-supplied out_col / out_row are returned untouched as there.
    """
    if A.dim() != 2:
        raise ValueError("Input must be 2D")
    _check_device(A, "double_quant")
    A = A.contiguous()
    rows, cols = A.shape
    dcode = dtype_code(A.dtype, "double_quant")
    cs = (torch.empty(cols, dtype=torch.float32, device=A.device) if col_stats is None
          else col_stats.to(device=A.device, dtype=torch.float32).contiguous().clone())
    rs = (torch.empty(rows, dtype=torch.float32, device=A.device) if row_stats is None
          else row_stats.to(device=A.device, dtype=torch.float32).contiguous().clone())
    oc = torch.empty(A.shape, dtype=torch.int8, device=A.device)
    orow = torch.empty(A.shape, dtype=torch.int8, device=A.device)
    with on_device(A.device):
        check(_native.lib().mbnb_double_quant(
            ptr(A), dcode, rows, cols, ptr(oc), ptr(orow), ptr(cs), ptr(rs), int(col_stats is not None),
            int(row_stats is not None), stream_ptr(A.device)), "double_quant")
    return (oc if out_col is None else out_col, orow if out_row is None else out_row,
            cs if col_stats is None else col_stats, rs if row_stats is None else row_stats, None)


def dequant_absmax(absmax_quant: Tensor, absmax_scales, blocksize: int = 256) -> Tensor:
    """Dequantize double-quantized absmax values (reference: functional.py:866-889).

    `absmax_scales` a QuantState: dequantize_blockwise (the live form, functional.py:870-871).  Otherwise the legacy
    form (functional.py:873-889): codes [rows, num_blocks] (or 1-D) with one scale per `blocksize` codes of a row;
    returns f32 ``codes.float() * scale``, zero where no scale block covers a code (the reference's zeros_like)."""
    if isinstance(absmax_scales, QuantState):
        return dequantize_blockwise(absmax_quant, absmax_scales)
    _check_device(absmax_quant, "dequant_absmax")
    q = absmax_quant.contiguous()
    rows = q.shape[0] if q.dim() > 1 else 1
    num_blocks = q.numel() // rows if rows > 0 else q.numel()
    scales = absmax_scales.to(device=q.device, dtype=torch.float32).contiguous()
    dq_blocks = scales.numel() // rows if rows > 0 else scales.numel()
    if blocksize <= 0:
        raise ValueError(f"blocksize must be positive, got {blocksize}")
    kind = 0 if q.dtype == torch.int8 else 1 if q.dtype == torch.uint8 else 2
    if kind == 2 and q.dtype != torch.float32:
        q = q.float()       # the reference's `.float()` of the codes
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    with on_device(q.device):
        check(_native.lib().mbnb_dequant_absmax(ptr(q), kind, rows, num_blocks, ptr(scales), dq_blocks, int(blocksize),
                                                ptr(out), stream_ptr(q.device)), "dequant_absmax")
    return out
