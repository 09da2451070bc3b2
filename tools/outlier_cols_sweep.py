#!/usr/bin/env python3
"""OutlierAwareLinear forward (mbnb_outlier_linear) at M = N = K = 4096, fp16, bias, by number of outlier columns: device
time per forward (HIP events, median of 5 x 50) and the int8 kernel that carried it.
    PYTHONPATH=. python tools/outlier_cols_sweep.py"""
import numpy as np
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic

dev = torch.device("cuda:0")
M = N = K = 4096
dt = torch.float16
x = synthetic.normal((M, K), dt, seed=824).to(dev)
b = synthetic.normal((N,), dt, seed=823).to(dev)
for n_out in (0, 8, 16, 32, 40, 64, 72, 128):
    W = synthetic.normal((N, K), torch.float32, seed=821, std=0.05)
    oidx = torch.from_numpy(np.sort(np.random.default_rng(5).choice(K, n_out, replace=False)).astype(np.int64))
    W[:, oidx] *= 30.0
    W = W.to(dt)
    W0 = W.clone()
    W0[:, oidx] = 0
    q, s = bnb.quantize_rowwise(W0.to(dev))
    ow = W[:, oidx].contiguous().to(dev)
    oi = oidx.to(dev)
    for _ in range(200):
        y = bnb.outlier_linear(x, q, s, oi, ow, b, dt)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = bnb.outlier_linear(x, q, s, oi, ow, b, dt)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1000)
    print(f"{n_out:4d} outlier columns: {sorted(ts)[2]:7.2f} us per forward ({_native.last_kernel()})")
