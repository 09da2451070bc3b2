"""
Quantized embedding tables on MI355X: Embedding4bit (NF4 / FP4), Embedding8bit (row-wise INT8), EmbeddingNF4/FP4.

Public surface of the reference module (mps_bitsandbytes/nn/embedding.py): constructor arguments, buffers
(`weight_packed` + `weight_absmax`; `weight_int8` + `weight_scales`), `from_embedding`, padding rows zeroed, results
bit-identical to its Python path.  A lookup is one gather + dequantize kernel (`functional.embedding_4bit/_8bit`).
"""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F
from ._base import sixteen_bit_or_half, source_device


class _QuantizedEmbedding(nn.Module):
    def _init_table(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int], dtype) -> None:
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx, self.dtype = padding_idx, dtype

    def _table_repr(self) -> str:
        return f'{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}'


class Embedding4bit(_QuantizedEmbedding):
    """Rows of `weight_packed` [num, dim/2] u8 with per-block absmax `weight_absmax` [num, ceil(dim/blocksize)] f32
    (reference nn/embedding.py:20-199)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None,
                 quant_type: str = 'nf4', blocksize: int = 64, device=None, dtype=torch.float16):
        super().__init__()
        if quant_type not in ('nf4', 'fp4'):
            raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type}")
        if embedding_dim % 2:
            raise ValueError(f"embedding_dim must be even, got {embedding_dim}")
        self._init_table(num_embeddings, embedding_dim, padding_idx, dtype)
        self.quant_type, self.blocksize = quant_type, blocksize
        blocks_per_row = -(-embedding_dim // blocksize)
        self.register_buffer('weight_packed', torch.zeros(num_embeddings, embedding_dim // 2, dtype=torch.uint8, device=device))
        self.register_buffer('weight_absmax', torch.ones(num_embeddings, blocks_per_row, dtype=torch.float32, device=device))

    def forward(self, input: Tensor) -> Tensor:
        """indices [*] -> [*, embedding_dim] in self.dtype."""
        return F.embedding_4bit(input, self.weight_packed, self.weight_absmax, self.embedding_dim, self.blocksize,
                                self.quant_type, self.padding_idx, self.dtype)

    @classmethod
    def from_embedding(cls, embedding: nn.Embedding, quant_type: str = 'nf4', blocksize: int = 64,
                       device=None) -> 'Embedding4bit':
        """Quantize an nn.Embedding (reference :140-193).  The reference quantises row by row; whenever its buffers can
        hold the result (embedding_dim a multiple of blocksize) that equals the 2-D row-wise quantiser — one launch here."""
        device = source_device(embedding.weight, device)
        table, dim = embedding.weight.data, embedding.embedding_dim
        if dim % 2:   # odd widths get one zero column, as in the reference
            table, dim = torch.nn.functional.pad(table, (0, 1)), dim + 1
        if dim % blocksize:
            raise ValueError(f"embedding_dim ({dim}) must be a multiple of blocksize ({blocksize}): the "
                             f"[num_embeddings, embedding_dim // 2] weight buffer cannot hold padded rows")
        layer = cls(embedding.num_embeddings, dim, padding_idx=embedding.padding_idx, quant_type=quant_type,
                    blocksize=blocksize, device=device, dtype=sixteen_bit_or_half(embedding.weight.dtype))
        packed, state = F.quantize_4bit(table.to(device).contiguous(), blocksize=blocksize, quant_type=quant_type)
        layer.weight_packed.copy_(packed.view_as(layer.weight_packed))
        layer.weight_absmax.copy_(state.absmax.view_as(layer.weight_absmax))
        return layer

    def extra_repr(self) -> str:
        return f'{self._table_repr()}, quant_type={self.quant_type}, blocksize={self.blocksize}'


class Embedding8bit(_QuantizedEmbedding):
    """Rows of `weight_int8` [num, dim] with `weight_scales` [num] f32 = row absmax (reference nn/embedding.py:202-303)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None, device=None,
                 dtype=torch.float16):
        super().__init__()
        self._init_table(num_embeddings, embedding_dim, padding_idx, dtype)
        self.register_buffer('weight_int8', torch.zeros(num_embeddings, embedding_dim, dtype=torch.int8, device=device))
        self.register_buffer('weight_scales', torch.ones(num_embeddings, dtype=torch.float32, device=device))

    def forward(self, input: Tensor) -> Tensor:
        return F.embedding_8bit(input, self.weight_int8, self.weight_scales, self.padding_idx, self.dtype)

    @classmethod
    def from_embedding(cls, embedding: nn.Embedding, device=None) -> 'Embedding8bit':
        device = source_device(embedding.weight, device)
        layer = cls(embedding.num_embeddings, embedding.embedding_dim, padding_idx=embedding.padding_idx,
                    device=device, dtype=sixteen_bit_or_half(embedding.weight.dtype))
        q, absmax = F.quantize_rowwise(embedding.weight.data.to(device))
        layer.weight_int8.copy_(q)
        layer.weight_scales.copy_(absmax)
        return layer

    def extra_repr(self) -> str:
        return self._table_repr()


def _fixed_table(name: str, qt: str):
    """Embedding4bit with the code table pinned (reference nn/embedding.py:307-328)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, **kwargs):
        kwargs['quant_type'] = qt
        Embedding4bit.__init__(self, num_embeddings, embedding_dim, **kwargs)

    def from_embedding(cls, embedding, blocksize: int = 64, device=None):
        return Embedding4bit.from_embedding.__func__(cls, embedding, quant_type=qt, blocksize=blocksize, device=device)

    return type(name, (Embedding4bit,), {'__init__': __init__, 'from_embedding': classmethod(from_embedding),
                                          '__doc__': f"Embedding4bit with quant_type='{qt}'.", '__module__': __name__})


EmbeddingNF4 = _fixed_table('EmbeddingNF4', 'nf4')
EmbeddingFP4 = _fixed_table('EmbeddingFP4', 'fp4')
