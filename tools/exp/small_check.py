#!/usr/bin/env python3
"""k_gemm_small against the older mid-sized-batch kernels: device time per call from a HIP graph of 20 calls (no host overhead).

    python tools/exp/small_check.py
"""
import statistics
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mps_bitsandbytes_amd as bnb  # noqa: E402
from mps_bitsandbytes_amd import _native  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


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
            e0.record(s)
            g.replay()
            e1.record(s)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


shapes = [(280, 4096, 4096), (320, 4096, 4096), (360, 4096, 4096), (300, 2048, 8192), (17, 4096, 4096), (24, 4096, 4096), (32, 4096, 4096), (32, 11008, 4096), (33, 4096, 4096), (48, 4096, 4096), (64, 4096, 4096), (64, 11008, 4096), (72, 4096, 4096), (96, 4096, 4096), (128, 4096, 4096), (192, 4096, 4096), (256, 4096, 4096), (128, 11008, 4096), (256, 11008, 4096),
          (128, 4096, 11008), (100, 1024, 1024), (200, 4096, 1024)]
for (M, N, K) in shapes:
    for dt, cs in ((torch.bfloat16, False), (torch.bfloat16, True)):
        g = torch.Generator(device=dev)
        g.manual_seed(M + N)
        W = torch.randn(N, K, generator=g, device=dev).to(dt)
        X = torch.randn(M, K, generator=g, device=dev).to(dt)
        packed, st = bnb.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=cs)
        y = bnb.matmul_4bit(X, packed, st)
        kern = _native.last_kernel()
        Wd = bnb.dequantize_4bit(packed, st)
        err = rel(y, X.float() @ Wd.float().t())
        us = graph_us(lambda: bnb.matmul_4bit(X, packed, st))
        # the same shape with blocksize 128 takes the older kernels (gemm_small needs blocksize 64): a timing reference
        p2, s2 = bnb.quantize_4bit(W, blocksize=128, quant_type="nf4", compress_statistics=cs)
        bnb.matmul_4bit(X, p2, s2)
        k2 = _native.last_kernel()
        us2 = graph_us(lambda: bnb.matmul_4bit(X, p2, s2))
        print(f"{M:4d} x {N:5d} x {K:5d} dq={int(cs)}  {kern:18s} {us:6.1f} us  (rel {err:.1e})   | blocksize 128 -> {k2:16s} {us2:6.1f} us", flush=True)
