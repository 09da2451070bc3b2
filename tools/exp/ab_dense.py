#!/usr/bin/env python3
"""Decode-once A/B: fused k_gemm256s (library matmul_4bit) against dequantize_4bit + the dense k_gemm256d variants
(tools/exp/gemm256d.h) and against dequantize_4bit + the vendor BLAS, interleaved in one process.

    python tools/exp/ab_dense.py [--rounds 7] [--iters 40] [--shape M,N,K]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mps_bitsandbytes_amd as bnb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,1")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--shapes", default="4096,4096,4096;2048,4096,4096;1024,4096,4096;8192,4096,4096;4096,11008,4096")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_exp.so"))
lib.exp_gemm256d.restype = ctypes.c_int
lib.exp_gemm256d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                             ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream

for shp in args.shapes.split(";"):
    M, N, K = [int(v) for v in shp.split(",")]
    g = torch.Generator(device=dev)
    g.manual_seed(M + N)
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    packed, state = bnb.quantize_nf4(W, blocksize=64)
    X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    ref = bnb.matmul_4bit(X, packed, state)
    Wd = bnb.dequantize_4bit(packed, state)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def dense(v):
        rc = lib.exp_gemm256d(v, X.data_ptr(), Wd.data_ptr(), out.data_ptr(), M, N, K, st)
        assert rc == 0, rc

    legs = {"fused": lambda: bnb.matmul_4bit(X, packed, state),
            "dequant": lambda: bnb.dequantize_4bit(packed, state, out=Wd),
            "blas": lambda: torch.matmul(X, Wd.t(), out=out)}
    for v in variants:
        out.fill_(float("nan"))
        dense(v)
        torch.cuda.synchronize()
        rel = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
        print(f"check {shp} dense variant {v}: {'bit-equal to the fused kernel' if torch.equal(out, ref) else f'DIFFERS (rel {rel:.2e})'}", flush=True)
        legs[f"dense{v}"] = (lambda v=v: dense(v))
        if v < 2:
            legs[f"dequant+dense{v}"] = (lambda v=v: (bnb.dequantize_4bit(packed, state, out=Wd), dense(v)))
    for f in legs.values():
        for _ in range(100):
            f()
    torch.cuda.synchronize()
    times = {k: [] for k in legs}
    for r in range(args.rounds):
        for k, f in legs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                f()
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            e1.synchronize()
            times[k].append(e0.elapsed_time(e1) / args.iters * 1e3)
    flops = 2.0 * M * N * K
    print(f"shape {M}x{N}x{K}: {args.rounds} interleaved rounds x {args.iters} launches (us per call)")
    for k, t in times.items():
        med = statistics.median(t)
        print(f"  {k:16s} median {med:8.2f}  min {min(t):8.2f}   {flops / med / 1e6:7.1f} TFLOP/s", flush=True)
