"""k_gemm_i8_inplace with the four-part epilogue (tools/exp/libi8_stamps.so, built from the working tree) against the installed library's matmul_int8 at 4096^3: same bits,
interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libi8_stamps.so"))
lib.exp_i8.restype = ctypes.c_int; lib.exp_i8.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8); B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
sA = torch.rand(M, generator=g, device=dev) + 0.5; sB = torch.rand(N, generator=g, device=dev) + 0.5
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
ref = bnb.matmul_int8(A, B, sA, sB, torch.bfloat16)
print("library kernel:", _native.last_kernel())
def new():
    assert lib.exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), out.data_ptr(), M, N, K, sp) == 0
def old():
    bnb.matmul_int8(A, B, sA, sB, torch.bfloat16)
new(); torch.cuda.synchronize()
print("equal:", torch.equal(out, ref))
def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for f in (old, new):
    for _ in range(50):
        f()
ev(old, 3000)
res = {"library (two-part epilogue)": [], "four-part epilogue, packed products": []}
for rep in range(9):
    res["library (two-part epilogue)"].append(ev(old, 300))
    res["four-part epilogue, packed products"].append(ev(new, 300))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:38s} median {v[4]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}")
