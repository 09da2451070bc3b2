"""384 < M <= 512 (and 256 < M <= 384): k_gemm_small with 16 steps of weights in registers (one K slice, one round of workgroups; round 3)
against the decode-once path (dequantize_4bit + k_gemm_dense128): device time per call from a HIP graph of 20 calls, and the two
results against each other (same Wd bits; different summation order -> tolerance, not equality)."""
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


for (M, N, K, cs) in [(257, 4096, 4096, False), (288, 4096, 4096, False), (320, 4096, 4096, False), (384, 4096, 4096, False), (448, 4096, 4096, False), (512, 4096, 4096, False),
                      (512, 4096, 4096, True), (512, 4096, 2048, False), (512, 2048, 4096, False), (400, 5120, 4096, False), (512, 11008, 4096, True), (512, 4096, 8192, False),
                      (640, 4096, 4096, False), (512, 2048, 8192, False), (384, 11008, 4096, False), (300, 1000, 3072, False), (512, 1024, 4096, False), (300, 4096, 1024, False),
                      (512, 8192, 2048, False), (257, 2048, 2048, False), (512, 8192, 4096, False), (384, 8192, 4096, False), (300, 6144, 4096, False), (512, 7168, 2048, False), (512, 8192, 8192, False), (400, 5120, 5120, False)]:
    g = torch.Generator(device=dev); g.manual_seed(M + N)
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16) * (0.05 if cs else 1.0)
    X = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=cs)
    res = {}
    for once in (True, False):
        F.DECODE_ONCE = once
        y = bnb.matmul_4bit(X, packed, st)
        res[once] = (y, _native.last_kernel(), graph_us(lambda: bnb.matmul_4bit(X, packed, st)))
    F.DECODE_ONCE = True
    d = ((res[True][0].double() - res[False][0].double()).norm() / res[True][0].double().norm()).item()
    print(f"{M:4d} x {N:5d} x {K:5d}{' dq' if cs else '   '}: decode-once {res[True][1]:18s} {res[True][2]:7.2f} us   fused {res[False][1]:18s} {res[False][2]:7.2f} us   rel diff {d:.2e}", flush=True)
