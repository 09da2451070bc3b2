"""Generic kernel at the shapes it still serves in practice: f32 weight dtype at M <= 4 (decode with an fp32 model)."""
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
N = K = 4096
for dt, bs in ((torch.float32, 64), (torch.bfloat16, 16)):
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(dt)
    p, st = bnb.quantize_nf4(W, blocksize=bs)
    for M in (1, 2, 4, 8):
        X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(dt)
        for _ in range(3):
            bnb.matmul_4bit(X, p, st)
        torch.cuda.synchronize()
        us = min(ev(lambda: bnb.matmul_4bit(X, p, st), 20) for _ in range(3))
        print("%s bs=%d M=%d: %.1f us (%s)" % (dt, bs, M, us, _native.last_kernel()))
