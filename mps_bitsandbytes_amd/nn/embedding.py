"""
Quantized embedding layers (reference: mps_bitsandbytes/nn/embedding.py).

Same classes, constructor arguments, buffers (`weight_packed`, `weight_absmax`; `weight_int8`, `weight_scales`)
and results as the reference; the lookup is ONE gather+dequantize HIP kernel per call
(`functional.embedding_4bit` / `embedding_8bit`, C ABI `mbnb_embedding_4bit/_8bit`), bit-exact against the
reference's Python path.  ROCm (`cuda`) tensors only — there is no CPU path.
"""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F


class Embedding4bit(nn.Module):
    """4-bit (NF4 / FP4) embedding table (reference: nn/embedding.py:20-199)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None,
                 quant_type: str = 'nf4', blocksize: int = 64, device=None, dtype=torch.float16):
        super().__init__()
        if quant_type not in ('nf4', 'fp4'):
            raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
        if embedding_dim % 2 != 0:
            raise ValueError(f"embedding_dim must be even, got {embedding_dim}")
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx
        self.quant_type = quant_type
        self.blocksize = blocksize
        self.dtype = dtype
        num_blocks = (embedding_dim + blocksize - 1) // blocksize
        self.register_buffer('weight_packed',
                             torch.zeros(num_embeddings, embedding_dim // 2, dtype=torch.uint8, device=device))
        self.register_buffer('weight_absmax',
                             torch.ones(num_embeddings, num_blocks, dtype=torch.float32, device=device))

    def forward(self, input: Tensor) -> Tensor:
        """input: indices [*]; returns [*, embedding_dim] in self.dtype (rows equal to padding_idx are zeros)."""
        return F.embedding_4bit(input, self.weight_packed, self.weight_absmax, self.embedding_dim, self.blocksize,
                                self.quant_type, self.padding_idx, self.dtype)

    @classmethod
    def from_embedding(cls, embedding: nn.Embedding, quant_type: str = 'nf4', blocksize: int = 64,
                       device=None) -> 'Embedding4bit':
        """Quantize an nn.Embedding (reference: nn/embedding.py:140-193).  The reference quantises row by row; for
        embedding_dim % blocksize == 0 — the only case its buffers can hold — that equals the 2-D row-wise
        quantiser, which is what runs here (one HIP launch)."""
        if device is None:
            device = embedding.weight.device
        dtype = embedding.weight.dtype
        if dtype not in (torch.float16, torch.bfloat16):
            dtype = torch.float16
        embedding_dim = embedding.embedding_dim
        weight = embedding.weight.data
        if embedding_dim % 2 != 0:
            embedding_dim = embedding_dim + 1
            weight = torch.nn.functional.pad(weight, (0, 1))
        if embedding_dim % blocksize != 0:
            raise ValueError(f"embedding_dim ({embedding_dim}) must be a multiple of blocksize ({blocksize}): the "
                             f"[num_embeddings, embedding_dim // 2] weight buffer cannot hold padded rows")
        layer = cls(embedding.num_embeddings, embedding_dim, padding_idx=embedding.padding_idx,
                    quant_type=quant_type, blocksize=blocksize, device=device, dtype=dtype)
        packed, state = F.quantize_4bit(weight.to(device).contiguous(), blocksize=blocksize, quant_type=quant_type)
        layer.weight_packed.copy_(packed.view(embedding.num_embeddings, embedding_dim // 2))
        layer.weight_absmax.copy_(state.absmax.view(embedding.num_embeddings, -1))
        return layer

    def extra_repr(self) -> str:
        return (f'{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}, '
                f'quant_type={self.quant_type}, blocksize={self.blocksize}')


class Embedding8bit(nn.Module):
    """Row-wise INT8 embedding table (reference: nn/embedding.py:202-303)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None, device=None,
                 dtype=torch.float16):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx
        self.dtype = dtype
        self.register_buffer('weight_int8', torch.zeros(num_embeddings, embedding_dim, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(num_embeddings, dtype=torch.float32, device=device))

    def forward(self, input: Tensor) -> Tensor:
        return F.embedding_8bit(input, self.weight_int8, self.weight_scales, self.padding_idx, self.dtype)

    @classmethod
    def from_embedding(cls, embedding: nn.Embedding, device=None) -> 'Embedding8bit':
        if device is None:
            device = embedding.weight.device
        dtype = embedding.weight.dtype
        if dtype not in (torch.float16, torch.bfloat16):
            dtype = torch.float16
        layer = cls(embedding.num_embeddings, embedding.embedding_dim, padding_idx=embedding.padding_idx,
                    device=device, dtype=dtype)
        weight_int8, weight_scales = F.quantize_rowwise(embedding.weight.data.to(device))
        layer.weight_int8.copy_(weight_int8)
        layer.weight_scales.copy_(weight_scales)
        return layer

    def extra_repr(self) -> str:
        return f'{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}'


class EmbeddingNF4(Embedding4bit):
    """Embedding4bit with quant_type='nf4' (reference: nn/embedding.py:307-316)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, **kwargs):
        kwargs['quant_type'] = 'nf4'
        super().__init__(num_embeddings, embedding_dim, **kwargs)

    @classmethod
    def from_embedding(cls, embedding, blocksize: int = 64, device=None) -> 'EmbeddingNF4':
        return super().from_embedding(embedding, quant_type='nf4', blocksize=blocksize, device=device)


class EmbeddingFP4(Embedding4bit):
    """Embedding4bit with quant_type='fp4' (reference: nn/embedding.py:319-328)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, **kwargs):
        kwargs['quant_type'] = 'fp4'
        super().__init__(num_embeddings, embedding_dim, **kwargs)

    @classmethod
    def from_embedding(cls, embedding, blocksize: int = 64, device=None) -> 'EmbeddingFP4':
        return super().from_embedding(embedding, quant_type='fp4', blocksize=blocksize, device=device)
