"""f32 weight dtype: generic kernel vs dequantise-once + k_gemm_f32 by row count (HIP graph of 8 calls: no host time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def graphed(fn):
    gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(8):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    return min(ev(gr.replay, 5) for _ in range(3)) / 8
for (N, K) in ((4096, 4096), (11008, 4096), (1024, 1024)):
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32)
    p, st = bnb.quantize_nf4(W, blocksize=64)
    for M in (1, 4, 5, 8, 16, 32, 64, 128):
        X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32)
        res = []
        for flag in (False, True):
            bnb.functional.DECODE_ONCE = flag
            us = graphed(lambda: bnb.matmul_4bit(X, p, st))
            res.append("%s %.1f us" % (_native.last_kernel(), us))
        bnb.functional.DECODE_ONCE = True
        print("N=%d K=%d M=%d: %s | %s" % (N, K, M, res[0], res[1]))
