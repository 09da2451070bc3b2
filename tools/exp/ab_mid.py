#!/usr/bin/env python3
"""k_gemm_mid (tools/exp/mid_exp.hip) against the library's current path for mid-sized M: correctness (rel. Frobenius error
vs the library result, which the parity tests pin to the oracle) and time per call for forced slice counts.

    python tools/exp/ab_mid.py [--Ms 48,64,128,256,512,1024,2048] [--N 4096] [--K 4096] [--slices 1,2,4,8]
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
from mps_bitsandbytes_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--Ms", default="48,64,100,128,256,512,1024,2048")
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--slices", default="1,2,4,8")
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmid_exp.so"))
lib.exp_mid.restype = ctypes.c_int
lib.exp_mid.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N, K = args.N, args.K
g = torch.Generator(device=dev)
g.manual_seed(0)
W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
packed, state = bnb.quantize_nf4(W, blocksize=64)
absmax = state.absmax.contiguous()
ws = torch.empty(8 * 2048 * N * 4, dtype=torch.uint8, device=dev)


def timeit(fn, iters):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


for M in [int(v) for v in args.Ms.split(",")]:
    X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    ref = bnb.matmul_4bit(X, packed, state)
    kern = _native.last_kernel()
    # the library call through Python costs host time; time it inside a graph for a fair device-side number
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bnb.matmul_4bit(X, packed, state)
        with torch.cuda.graph(gr, stream=side):
            for _ in range(8):
                bnb.matmul_4bit(X, packed, state)
    torch.cuda.current_stream().wait_stream(side)
    t_lib = timeit(gr.replay, 10) / 8
    line = f"M={M:5d}  library {kern:16s} {t_lib:7.2f} us |"
    for s in [int(v) for v in args.slices.split(",")]:
        if s > K // 256:
            continue
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)

        def call():
            rc = lib.exp_mid(s, X.data_ptr(), packed.data_ptr(), absmax.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), M, N, K, K, st)
            assert rc == 0, rc
        call()
        torch.cuda.synchronize()
        err = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
        bad = not (err < 2e-3) or not bool(torch.isfinite(out).all())
        t = timeit(call, args.iters)
        line += f"  s={s}: {t:7.2f} us err {err:.1e}{' BAD' if bad else ''}"
    print(line, flush=True)
