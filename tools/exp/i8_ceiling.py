"""Bare MFMA loop ceilings of the box for bf16 and int8 (mbnb_probe_mfma kinds 0 / 1), next to the int8 dense GEMM, and
the cost of the generic (f32 weight dtype) matmul_4bit kernel at a mid-sized shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native

dev = torch.device("cuda:0")
import ctypes
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmbnb_probe.so"))   # make -C tools
lib.mbnb_probe_mfma.restype = ctypes.c_int64
lib.mbnb_probe_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(1, dtype=torch.float32, device=dev)
stp = torch.cuda.current_stream().cuda_stream


def probe(kind, ops_per):
    lib.mbnb_probe_mfma(kind, 2000, sink.data_ptr(), stp)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 0
        for _ in range(10):
            n += int(lib.mbnb_probe_mfma(kind, 20000, sink.data_ptr(), stp))
        e1.record(); e1.synchronize()
        best = max(best, n * ops_per / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("bare bf16 32x32x16 loop: %.1f TFLOP/s" % probe(0, 32768.0))
print("bare i8 32x32x32 loop:   %.1f TOP/s" % probe(1, 65536.0))
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randint(-127, 128, (4096, 4096), generator=g, device=dev, dtype=torch.int8)
B = torch.randint(-127, 128, (4096, 4096), generator=g, device=dev, dtype=torch.int8)
sa = torch.rand(4096, generator=g, device=dev) + 0.5
sb = torch.rand(4096, generator=g, device=dev) + 0.5
for _ in range(5):
    bnb.matmul_int8(A, B, sa, sb, dtype=torch.bfloat16)
us = min(ev(lambda: bnb.matmul_int8(A, B, sa, sb, dtype=torch.bfloat16), 50) for _ in range(3))
print("matmul_int8 4096^3: %.1f us = %.1f TOP/s (%s)" % (us, 2 * 4096**3 / us / 1e6, _native.last_kernel()))
# generic path: f32 weight dtype
for (M, N, K) in ((8, 4096, 4096), (64, 4096, 4096), (256, 4096, 4096), (1024, 4096, 4096), (4096, 4096, 4096)):
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32)
    p, st = bnb.quantize_nf4(W, blocksize=64)
    X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32)
    bnb.matmul_4bit(X, p, st)
    torch.cuda.synchronize()
    us = min(ev(lambda: bnb.matmul_4bit(X, p, st), 10) for _ in range(3))
    Wd = bnb.dequantize_4bit(p, st)
    torch.matmul(X, Wd.t())
    usb = ev(lambda: torch.matmul(X, Wd.t()), 10)
    print("f32 matmul_4bit M=%d: %.1f us (%s) = %.2f TFLOP/s; torch f32 matmul on the dequantised weight %.1f us" % (M, us, _native.last_kernel(), 2.0 * M * N * K / us / 1e6, usb))
