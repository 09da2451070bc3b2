"""M = 1 on LLM layer shapes beyond 4096^2: device time per call (HIP graph over 8 distinct layers of the shape, rotating: ~HBM-resident for the
larger ones), kernel taken, and the fraction of 8 TB/s for the layer's packed bytes + absmax."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")


def graph_us(fn, n, reps=7):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        ts = []
        for _ in range(reps):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


for (N, K, dq) in [(4096, 4096, False), (11008, 4096, False), (4096, 11008, False), (11008, 4096, True), (14336, 4096, False), (4096, 14336, False), (5120, 5120, False),
                   (8192, 8192, False), (28672, 8192, False), (8192, 28672, False), (6144, 4096, False), (4096, 2048, False), (2048, 8192, False)]:
    L = 24 if N * K <= 64 << 20 else 8
    layers = []
    for i in range(L):
        W = torch.randn(N, K, device=dev).to(torch.bfloat16) * (0.05 if dq else 1.0)
        layers.append(bnb.quantize_nf4(W, blocksize=64, compress_statistics=dq)); del W
    x = torch.randn(1, K, device=dev).to(torch.bfloat16)
    bnb.matmul_4bit(x, *layers[0]); kern = _native.last_kernel()

    def run():
        for p, st in layers:
            bnb.matmul_4bit(x, p, st)
    t = graph_us(run, L)
    byts = N * K / 2 + N * K / 64 * (1 if dq else 4) + K * 2 + N * 2
    print(f"{N:6d} x {K:6d}{' dq' if dq else '   '}: {kern:8s} {t:7.2f} us per layer   {byts / t / 1e6:5.2f} TB/s = {byts / t / 8e6:.3f} of 8 TB/s", flush=True)
    del layers
