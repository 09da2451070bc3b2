"""GEMV (M<=16) bandwidth vs weight size: separates the launch floor from the streaming rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, K) in [(4096, 4096), (11008, 4096), (4096, 11008), (28672, 8192), (65536, 8192)]:
    for M in (1, 2, 4, 8, 16):
        W = torch.randn(N, K, device=dev, dtype=torch.float16)
        packed, st = bnb.quantize_nf4(W); del W
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            bnb.matmul_4bit(x, packed, st)
            with torch.cuda.graph(g, stream=s):
                for _ in range(8): bnb.matmul_4bit(x, packed, st)
        torch.cuda.current_stream().wait_stream(s)
        us = timeit(g.replay, 20) / 8
        nbytes = N * K // 2 + N * (K // 64) * 4
        print(f"N={N} K={K} M={M}: {us:.2f} us/launch, {nbytes/us/1e3:.0f} GB/s ({_native.last_kernel()}) weights {nbytes/1e6:.1f} MB", flush=True)
