#!/usr/bin/env python3
"""Per-kernel durations of the small-M path for rocprofv3 --kernel-trace --stats:  rocprofv3 ... -- python3 tools/exp/small_prof.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mps_bitsandbytes_amd as bnb  # noqa: E402

dev = torch.device("cuda:0")
W = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W, blocksize=64)
for M in (64, 128, 256):
    X = torch.randn(M, 4096, device=dev).to(torch.bfloat16)
    for _ in range(200):
        bnb.matmul_4bit(X, packed, st)
torch.cuda.synchronize()
