#!/usr/bin/env python3
"""k_gemm_small8 (W8A16, 32 < M <= 256): correctness against dequantize_rowwise + f32 matmul and device time per call (HIP graph of 20)."""
import os
import statistics
import sys

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


for (M, N, K) in [(33, 4096, 4096), (48, 4096, 4096), (64, 4096, 4096), (96, 4096, 4096), (128, 4096, 4096), (200, 1000, 1024), (256, 4096, 4096),
                  (128, 11008, 4096), (100, 520, 768)]:
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator(device=dev)
        g.manual_seed(M + N)
        W = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(dt)
        X = torch.randn(M, K, generator=g, device=dev).to(dt)
        b = torch.randn(N, generator=g, device=dev).to(dt)
        q, s = bnb.quantize_rowwise(W)
        y = bnb.linear_int8(X, q, s, b)
        kern = _native.last_kernel()
        Wd = bnb.dequantize_rowwise(q, s, dt)
        e1 = rel(y, X.float() @ Wd.float().t() + b.float())
        us = graph_us(lambda: bnb.linear_int8(X, q, s, b))
        q8, s8 = bnb.quantize_fp8_e4m3(W)
        y8 = bnb.matmul_fp8_e4m3(X, q8, s8, b, dt)
        k8 = _native.last_kernel()
        Wd8 = bnb.dequantize_fp8_e4m3(q8, s8, dt)
        e8 = rel(y8, X.float() @ Wd8.float().t() + b.float())
        us8 = graph_us(lambda: bnb.matmul_fp8_e4m3(X, q8, s8, b, dt))
        print(f"{M:4d} x {N:5d} x {K:5d} {str(dt)[6:]:9s} {kern:20s} {us:6.1f} us (rel {e1:.1e})   {k8:20s} {us8:6.1f} us (rel {e8:.1e})", flush=True)
