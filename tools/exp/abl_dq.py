"""Timing-only ablation of k_gemm_dq (tools/exp/libdq_exp.so), interleaved with k_gemm_dense alone and the two-launch step.
ABL bits: 1 no side work in the loop, 2 no decode, 4 no stores, 8 no flag adds, 16 no polls, 32 no next-slab loads."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, functional as F
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdq_exp.so"))
lib.exp_dq.restype = ctypes.c_int
lib.exp_dq.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,4,24,32,62").split(",")]
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(1)
W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16)
x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W, blocksize=64)
Wd = bnb.dequantize_4bit(packed, st).contiguous()
sync = torch.zeros(32768, dtype=torch.uint8, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
nlib = _native.lib()
F.DECODE_IN_LAUNCH = False
ref = bnb.matmul_4bit(x, packed, st)

def dq(v):
    rc = lib.exp_dq(v, x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), Wd.data_ptr(), sync.data_ptr(), out.data_ptr(), M, N, K, sp)
    assert rc == 0, rc

def dense():
    rc = nlib.mbnb_gemm_dense(x.data_ptr(), Wd.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, None, 0, 1, sp)
    assert rc == 0, rc

def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dq(0); torch.cuda.synchronize()
print("variant 0 == two-launch:", torch.equal(out, ref), "sync nonzero:", int((sync != 0).sum()), flush=True)
legs = {"dense alone": dense, "two launches": lambda: bnb.matmul_4bit(x, packed, st)}
for v in variants:
    legs[f"dq abl={v}"] = (lambda v=v: (sync.zero_() if v & 8 else None, dq(v)))
for f in legs.values():
    for _ in range(20):
        f()
torch.cuda.synchronize()
ev(dense, 3000)
res = {k: [] for k in legs}
for rep in range(7):
    for k, f in legs.items():
        res[k].append(ev(f, 100))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:14s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)
