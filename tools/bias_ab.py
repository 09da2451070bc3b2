#!/usr/bin/env python3
"""matmul_4bit at M = N = K = 4096 (bf16, NF4) with and without a bias: device time per call (HIP events, median of 5 x 50).
    PYTHONPATH=. python tools/bias_ab.py            (MBNB_LIB=<path to another build of libmbnb_hip.so> for an A/B)"""
import os
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic

if os.environ.get("MBNB_LIB"):
    _native.LIB_PATH = os.environ["MBNB_LIB"]   # before the first call loads the library
    print("library:", _native.LIB_PATH)

dev = torch.device("cuda:0")
M = N = K = 4096
W = synthetic.normal((N, K), torch.bfloat16, seed=1, std=0.05).to(dev)
x = synthetic.normal((M, K), torch.bfloat16, seed=2).to(dev)
b = synthetic.normal((N,), torch.bfloat16, seed=3).to(dev)
q, st = bnb.quantize_4bit(W, quant_type="nf4")
for name, bias in (("no bias", None), ("bias", b)):
    for _ in range(300):
        y = bnb.matmul_4bit(x, q, st, bias=bias)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = bnb.matmul_4bit(x, q, st, bias=bias)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1000)
    ref = (x[:8].float() @ bnb.dequantize_4bit(q, st).float().t()) + (0 if bias is None else bias.float())
    err = ((y[:8].float() - ref).norm() / ref.norm()).item()
    print(f"{name}: {sorted(ts)[2]:.2f} us per call ({_native.last_kernel()}), rel err of 8 rows vs dequantised f32 product {err:.2e}")
