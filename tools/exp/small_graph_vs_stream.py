"""256 x 4096^2: the same k_gemm_small<4, 1, 16> launch timed four ways -- harness kernel / library call, back to back on a stream / from a HIP graph
of 20 -- to see what a graph replay adds per kernel node."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libm0_pk0.so"))
I64, P = ctypes.c_int64, ctypes.c_void_p
lib.exp_small_v.restype = ctypes.c_int; lib.exp_small_v.argtypes = [P] * 4 + [I64] * 3 + [P, ctypes.c_int]


def stream_us(fn, n=400):
    for _ in range(50):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


def graph_us(fn, n=20, reps=7):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn(s.cuda_stream)
        ts = []
        for _ in range(reps):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


g = torch.Generator(device=dev); g.manual_seed(5)
for (M, N, K, var) in [(256, 4096, 4096, 1), (512, 4096, 4096, 0)]:
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    cur = torch.cuda.current_stream().cuda_stream
    h = lambda sp=None: lib.exp_small_v(x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), out.data_ptr(), M, N, K, sp if sp is not None else cur, var)
    l = lambda sp=None: bnb.matmul_4bit(x, packed, st)
    l(); print(M, N, K, "library kernel:", _native.last_kernel())
    print(f"  harness kernel  stream {stream_us(h):6.2f} us   graph {graph_us(h):6.2f} us")
    print(f"  library call    stream {stream_us(l):6.2f} us   graph {graph_us(l):6.2f} us", flush=True)
