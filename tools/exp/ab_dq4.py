"""Variants of the large-matrix dequantize_4bit kernel (tools/exp/dq4_exp.hip) against the library's: bit equality, device time per call (HIP
graph of 20 calls into a fixed output), at 4096^2 and 11008 x 4096 bf16 NF4 blocksize 64; torch's fill and copy of the same output for scale."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdq4_exp.so"))
lib.exp_dq4.restype = ctypes.c_int
lib.exp_dq4.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
nlib = _native.lib()


def graph_us(fn, n=20, reps=9):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn(s)
        ts = []
        for _ in range(reps):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


for (N, K) in [(4096, 4096), (11008, 4096)]:
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    ref = bnb.dequantize_4bit(packed, st)
    out = torch.empty_like(ref)
    desc = None

    print(f"{N} x {K}: bytes moved {N * K * 2.53125 / 1e6:.1f} MB")
    def libcall(s):
        with torch.cuda.stream(s):
            bnb.dequantize_4bit(packed, st)
    print(f"  library k_dequantize_4bit           {graph_us(libcall):7.2f} us")
    for v in (1, 2, 4, 8, 12, 14, 18):
        out.fill_(float("nan"))
        def f(s, v=v):
            rc = lib.exp_dq4(v, packed.data_ptr(), st.absmax.data_ptr(), out.data_ptr(), N, K, s.cuda_stream)
            assert rc == 0, rc
        f(torch.cuda.current_stream()); torch.cuda.synchronize()
        print(f"  flat, {v % 10} dwords per thread{', nt stores' if v > 10 else '           '}  {graph_us(f):7.2f} us   equal: {torch.equal(out, ref)}", flush=True)
    print(f"  torch zero_ of the output           {graph_us(lambda s: out.zero_()):7.2f} us;   copy_ of it {graph_us(lambda s: out.copy_(ref)):7.2f} us")
