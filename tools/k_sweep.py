"""Kernel time of matmul_4bit at M=N=4096 for several K: separates per-k-step cost from prologue+epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
for K in (64, 128, 512, 2048, 4096, 8192):
    W = torch.randn(4096, K, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W)
    X = torch.randn(4096, K, device=dev).to(torch.bfloat16)
    for _ in range(10): bnb.matmul_4bit(X, packed, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): bnb.matmul_4bit(X, packed, st)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"K={K}: {us:.1f} us  ({2*4096*4096*K/us/1e6:.0f} TFLOP/s) {_native.last_kernel()}", flush=True)
