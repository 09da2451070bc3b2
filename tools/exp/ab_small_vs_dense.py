"""Shapes at the edge of the 257-512-row rule: the library's choice against dequantise + dense forced (dequantize_4bit into a scratch + functional.linear_dense), HIP graph."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, functional as F
dev = torch.device("cuda:0")


def graph_us(fn, n=20, reps=7):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        ts = []
        for _ in range(reps):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


for (M, N, K) in [(512, 8192, 4096), (384, 8192, 4096), (300, 6144, 4096), (512, 7168, 2048), (512, 4096, 8192), (512, 2048, 8192), (400, 5120, 4096), (512, 4096, 4096), (300, 4096, 4096)]:
    W = torch.randn(N, K, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    wd = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    y = bnb.matmul_4bit(x, packed, st); kern = _native.last_kernel()
    t_lib = graph_us(lambda: bnb.matmul_4bit(x, packed, st))
    def dense():
        bnb.dequantize_4bit(packed, st, out=wd)
        return F.linear_dense(x, wd)
    t_dense = graph_us(dense)
    print(f"{M:4d} x {N:5d} x {K:5d}: library {kern:18s} {t_lib:7.2f} us   dequantise + dense forced {t_dense:7.2f} us", flush=True)
