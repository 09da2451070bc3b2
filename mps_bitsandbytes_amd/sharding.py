"""
Batch-row sharding of a quantized linear across the GPUs of one node.

`out[m, :]` depends only on `A[m, :]` and the read-only weight, so the path shards by rows with
no data-path collective: weights (packed + absmax [+ state2] + bias) are replicated, each rank
computes its contiguous block of rows.  The one exchange step the north-star names is an
all-gather of the output shards (RCCL over xGMI on GPUs, `backend="nccl"`; gloo on CPU in the
tests).  One process per GPU; this module holds no kernel code.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def row_shard(M_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of the rows owned by `rank`: contiguous blocks, the first
    `M_global % world_size` ranks take one extra row."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, extra = divmod(M_global, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sharded_linear(local_rows: torch.Tensor, forward: Callable[[torch.Tensor], torch.Tensor], M_global: int,
                   gather: bool = True, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Apply `forward` (e.g. a Linear4bit) to this rank's rows and, if `gather`, all-gather the
    [M_global, N] result on every rank.  Equal shards use one all_gather_into_tensor (a single
    RCCL call on GPUs); ragged shards are padded to the largest shard for the same single call."""
    y_local = forward(local_rows)
    if not gather or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return y_local
    world = dist.get_world_size(group)
    N = y_local.shape[-1]
    if out is None:
        out = torch.empty(M_global, N, dtype=y_local.dtype, device=y_local.device)
    if M_global % world == 0:
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
    else:
        # ragged: pad every shard to the largest one, gather once, then drop the pad rows
        max_rows = (M_global + world - 1) // world
        padded = torch.zeros(max_rows, N, dtype=y_local.dtype, device=y_local.device)
        padded[: y_local.shape[0]] = y_local
        staged = torch.empty(world * max_rows, N, dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(staged, padded, group=group)
        for r in range(world):
            s, e = row_shard(M_global, r, world)
            out[s:e] = staged[r * max_rows: r * max_rows + (e - s)]
    return out


class ChunkedGather:
    """Row-chunked overlap of the output all-gather INSIDE one step (third curve).

    The rank's `rows` rows are cut into `chunks` contiguous chunks.  Chunk i is pushed through `forward` on the
    caller's stream and handed to an asynchronous `all_gather_into_tensor`, so the collective of chunk i (RCCL's own
    stream on GPUs) runs under the GEMM of chunk i + 1; only the last chunk's gather is exposed.  Every collective
    stays one contiguous [world, rows/c, N] slab; the result buffer is laid out [chunk][rank][row-in-chunk][N] and
    returned as a strided VIEW in global row order (rank, chunk, row) -- `.reshape(M_global, N)` materialises it.
    Two result buffers alternate between steps: a buffer is rewritten only after the gathers that filled it two steps
    earlier were waited for.  `rows` must be divisible by `chunks` (equal shards; ragged batches use sharded_linear)."""

    def __init__(self, forward: Callable[[torch.Tensor], torch.Tensor], rows: int, N: int, dtype: torch.dtype, device,
                 world_size: int, chunks: int = 2, group: Optional[dist.ProcessGroup] = None):
        if chunks < 1 or rows % chunks != 0:
            raise ValueError(f"ChunkedGather: {rows} rows are not divisible into {chunks} chunks")
        self.forward, self.rows, self.N, self.world, self.chunks, self.group = forward, rows, N, world_size, chunks, group
        self.rc = rows // chunks
        self.buffers = [torch.empty(chunks, world_size, self.rc, N, dtype=dtype, device=device) for _ in range(2)]
        self.pending = [[], []]
        self.count = 0

    def step(self, local_rows: torch.Tensor) -> torch.Tensor:
        b = self.count & 1
        self.count += 1
        self._drain(b)
        buf = self.buffers[b]
        for i in range(self.chunks):
            y = self.forward(local_rows[i * self.rc:(i + 1) * self.rc]).contiguous()
            work = dist.all_gather_into_tensor(buf[i].view(self.world * self.rc, self.N), y, group=self.group, async_op=True)
            # the chunk's shard stays referenced until its gather has been waited for: the collective reads it on another
            # stream, and dropping the last reference would hand its memory back to the allocator of THIS stream
            self.pending[b].append((work, y))
        return buf.permute(1, 0, 2, 3)   # [rank][chunk][row][N]: global row order; valid after finish()

    def _drain(self, b: int) -> None:
        for work, _shard in self.pending[b]:
            work.wait()
        self.pending[b] = []

    def finish(self) -> None:
        for b in range(2):
            self._drain(b)
