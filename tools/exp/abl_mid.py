#!/usr/bin/env python3
"""Timing-only ablation of k_gemm_mid (M=128, 4096x4096, slices 1 and 4): bit 1 no activation DMA, 2 no MFMA, 4 no decode
products / image write, 8 no raw / absmax DMA."""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mps_bitsandbytes_amd as bnb
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmid_exp.so"))
lib.exp_mid_abl.restype = ctypes.c_int
lib.exp_mid_abl.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 4096, 4096
g = torch.Generator(device=dev); g.manual_seed(0)
W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
packed, state = bnb.quantize_nf4(W, blocksize=64)
absmax = state.absmax.contiguous()
X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device=dev)
for s in (1, 4):
    for abl in (0, 1, 2, 3, 4, 6, 7, 9, 15):
        def call():
            rc = lib.exp_mid_abl(abl, s, X.data_ptr(), packed.data_ptr(), absmax.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), M, N, K, K, st)
            assert rc == 0
        for _ in range(20): call()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): call()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"M={M} slices={s} ablate={abl:2d}: {statistics.median(ts):7.2f} us", flush=True)
