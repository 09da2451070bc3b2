"""Cycles of k_gemm_i8_inplace's k-loop (32 k-steps of 128 v_mfma_i32_16x16x64_i8 per wave at 4096^3) and the kernel's time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libi8_stamps.so"))
lib.exp_i8.restype = ctypes.c_int; lib.exp_i8.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
lib.exp_i8_stamps.restype = ctypes.c_int; lib.exp_i8_stamps.argtypes = [ctypes.c_void_p]
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8); B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
sA = torch.rand(M, generator=g, device=dev) + 0.5; sB = torch.rand(N, generator=g, device=dev) + 0.5
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    assert lib.exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), out.data_ptr(), M, N, K, sp) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    lib.exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), out.data_ptr(), M, N, K, sp)
e1.record(); e1.synchronize()
host = (ctypes.c_ulonglong * (8 + 4 * 256))()
assert lib.exp_i8_stamps(host) == 0
us = e0.elapsed_time(e1) / 200 * 1e3
for wv in range(4):
    print(f"wave {wv}: k-loop {host[wv]} cycles = {host[wv] / 32:.0f} per k-step of 128 MFMAs; kernel {us:.2f} us per call")
import statistics
t = [[host[8 + 4 * b + i] for i in range(4)] for b in range(256)]
t0 = min(r[0] for r in t)
for name, i in (("start", 0), ("loop begins", 1), ("loop ends", 2), ("end", 3)):
    v = sorted((r[i] - t0) / 100 for r in t)
    print(f"{name:12s}: first {v[0]:6.2f}  median {v[128]:6.2f}  last {v[-1]:6.2f} us")
