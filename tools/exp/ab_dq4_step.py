"""The whole large-M step with the nontemporal-store dequantise kernel (tools/exp/dq4_exp.hip variant v) + the library's dense GEMM against the
library's matmul_4bit (k_dequantize_4bit + k_gemm_dense): same bits, interleaved timing (stream launches, 200 calls per sample)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdq4_exp.so"))
lib.exp_dq4.restype = ctypes.c_int
lib.exp_dq4.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
nlib = _native.lib()
SHAPES = [(4096, 4096, 4096), (4096, 11008, 4096), (1024, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "--sweep":
    SHAPES = [(2048, 4096, 4096), (8192, 4096, 4096), (4096, 2048, 2048), (4096, 8192, 4096), (4096, 4096, 11008), (2048, 8192, 8192), (1536, 4096, 4096), (600, 4096, 4096), (1024, 11008, 4096), (1024, 2048, 2048), (768, 5120, 5120)]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(7)
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    ref = bnb.matmul_4bit(x, packed, st)
    wd = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws_b = int(nlib.mbnb_gemm_dense_workspace_bytes(M, N, K)); ws = torch.empty(max(ws_b, 16), dtype=torch.uint8, device=dev)

    def step(v):
        def f():
            rc = lib.exp_dq4(v, packed.data_ptr(), st.absmax.data_ptr(), wd.data_ptr(), N, K, sp); assert rc == 0, rc
            rc = nlib.mbnb_gemm_dense(x.data_ptr(), wd.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, ws.data_ptr(), ws.numel(), 0, sp); assert rc == 0, rc
        return f
    only = len(sys.argv) > 1 and sys.argv[1] == "--sweep"
    legs = {"library matmul_4bit": lambda: bnb.matmul_4bit(x, packed, st), "flat 1 dword sc1 + dense": step(22), "flat 4 dwords sc1 + dense": step(34), "flat 3 dwords sc1 + dense": step(33), "flat 5 dwords sc1 + dense": step(35), "xcd-contiguous 4 dwords sc1": step(54), "xcd-contiguous 1 dword sc1": step(51)}
    if only:
        legs = {k: legs[k] for k in ("library matmul_4bit", "flat 1 dword + dense", "flat 1 dword sc1 + dense", "flat 4 dwords + dense", "flat 4 dwords sc1 + dense")}
    for name, f in legs.items():
        f(); torch.cuda.synchronize()
        if name != "library matmul_4bit":
            print(f"{M} x {N} x {K}  {name}: equal to the library's result: {torch.equal(out, ref)}", flush=True)
    for f in legs.values():
        for _ in range(30):
            f()
    ev(legs["library matmul_4bit"], 1500)
    res = {k: [] for k in legs}
    for rep in range(7):
        for k, f in legs.items():
            res[k].append(ev(f, 200))
    for k, v in res.items():
        v = sorted(v)
        print(f"  {k:28s} median {v[3]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)
