"""matmul_4bit time vs M at 4096x4096 NF4 bf16 (which kernel serves which batch size, and how far from the M = 1 stream time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
N = K = 4096
W = torch.randn(N, K, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W); del W
for M in (1, 2, 4, 8, 16, 17, 24, 32, 48, 64, 128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bnb.matmul_4bit(x, packed, st)
        with torch.cuda.graph(g, stream=s):
            for _ in range(8): bnb.matmul_4bit(x, packed, st)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 / 8 * 1e3
    print(f"M={M}: {us:.2f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s  ({_native.last_kernel()})", flush=True)
