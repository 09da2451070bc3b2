#!/usr/bin/env python3
"""One vendor-BLAS bf16 GEMM (X [M, K] @ Wd [N, K]^T) for rocprofv3 --kernel-trace: which library kernel runs, how long."""
import torch
M = N = K = 4096
X = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(50):
    torch.matmul(X, W.t(), out=out)
torch.cuda.synchronize()
