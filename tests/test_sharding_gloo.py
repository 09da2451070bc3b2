"""The N>1 path (batch-row sharding + all-gather of the output shards) on CPU with the gloo
backend, world_size 2 and 3 (ragged).  The per-rank "forward" here is the CPU oracle (tests may
use it); on the GPU box the same sharding code wraps the HIP Linear4bit (bench.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, return_dict):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from mps_bitsandbytes_amd import synthetic
        from mps_bitsandbytes_amd.sharding import row_shard, sharded_linear
        torch.set_num_threads(1)
        oracle.set_num_threads(1)
        N, K = 48, 128
        W = synthetic.normal((N, K), torch.float16, seed=5)           # replicated weight
        packed, absmax, _ = oracle.quantize_4bit(W, 64, "nf4")
        X = synthetic.normal((M, K), torch.float16, seed=6)           # global batch
        s, e = row_shard(M, rank, world)

        def fwd(x):
            return oracle.matmul_4bit(x, packed, absmax, (N, K), 64, "nf4", torch.float16)

        full = sharded_linear(X[s:e], fwd, M, gather=True)
        ref = fwd(X)
        return_dict[rank] = bool(torch.equal(full, ref)) and tuple(full.shape) == (M, N)
        local = sharded_linear(X[s:e], fwd, M, gather=False)
        return_dict[f"local{rank}"] = bool(torch.equal(local, ref[s:e]))
        if M % world == 0 and (M // world) % 2 == 0:
            # row-chunked overlap inside one step: two steps through the two alternating buffers
            from mps_bitsandbytes_amd.sharding import ChunkedGather
            cg = ChunkedGather(fwd, e - s, N, torch.float16, "cpu", world, chunks=2)
            ok = True
            for _ in range(3):
                view = cg.step(X[s:e])
                # round 4: until its gather has been waited for, every chunk's shard stays referenced next to its work handle
                held = [p for b in cg.pending for p in b]
                ok = ok and len(held) == 2 and all(isinstance(p, tuple) and p[1].shape == ((e - s) // 2, N) and p[1].is_contiguous() for p in held)
                cg.finish()
                ok = ok and cg.pending == [[], []]
                ok = ok and bool(torch.equal(view.reshape(M, N), ref))
            # two steps in flight: the buffer of step i is only rewritten after step i's gathers were waited for
            v1 = cg.step(X[s:e]); v2 = cg.step(X[s:e])
            ok = ok and sum(len(b) for b in cg.pending) == 4
            cg.finish()
            ok = ok and bool(torch.equal(v1.reshape(M, N), ref)) and bool(torch.equal(v2.reshape(M, N), ref))
            return_dict[f"chunked{rank}"] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M", [(2, 64), (3, 50)])
def test_row_sharded_linear_allgather(world, M):
    import oracle
    oracle.lib()  # build once in the parent
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        assert ret[r] is True and ret[f"local{r}"] is True
        if M % world == 0 and (M // world) % 2 == 0:
            assert ret[f"chunked{r}"] is True
