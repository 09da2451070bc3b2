"""k_gemm_dense reading the weight operand from a TILE-MAJOR scratch (libdense_tm1.so: each LDS-DMA piece 1 KiB contiguous) against the
row-major scratch (libdense_tm0.so): bit equality, then device time per launch (HIP graph of 10, 200 ms of load first, median of 9, alternating)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
libs = [ctypes.CDLL(os.path.join(here, f"libdense_tm{v}.so")) for v in (0, 1)]
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in libs:
    l.exp_dense.restype = ctypes.c_int; l.exp_dense.argtypes = [P] * 3 + [I64] * 3 + [P]
    l.exp_relayout.restype = ctypes.c_int; l.exp_relayout.argtypes = [P, P, I64, I64, P]
assert libs[0].exp_tile_major() == 0 and libs[1].exp_tile_major() == 1


def graph(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    return g


def t_us(g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


for (M, N, K) in [(4096, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 11008), (8192, 4096, 4096), (4096, 12288, 4096), (3000, 5000, 2048)]:
    gen = torch.Generator(device=dev); gen.manual_seed(M + N + K)
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=gen, device=dev) * 0.05).to(torch.bfloat16)
    wt = torch.zeros(((N + 255) // 256) * 256 * K, dtype=torch.bfloat16, device=dev)
    sp = lambda: torch.cuda.current_stream().cuda_stream
    assert libs[1].exp_relayout(w.data_ptr(), wt.data_ptr(), N, K, sp()) == 0
    o0 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    o1 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    assert libs[0].exp_dense(x.data_ptr(), w.data_ptr(), o0.data_ptr(), M, N, K, sp()) == 0
    assert libs[1].exp_dense(x.data_ptr(), wt.data_ptr(), o1.data_ptr(), M, N, K, sp()) == 0
    torch.cuda.synchronize()
    eq = torch.equal(o0, o1) and bool(torch.isfinite(o1.float()).all())
    g0 = graph(lambda: libs[0].exp_dense(x.data_ptr(), w.data_ptr(), o0.data_ptr(), M, N, K, sp()))
    g1 = graph(lambda: libs[1].exp_dense(x.data_ptr(), wt.data_ptr(), o1.data_ptr(), M, N, K, sp()))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        g0.replay(); g1.replay()
        torch.cuda.synchronize()
    a, b = [], []
    for _ in range(9):
        a.append(t_us(g0)); b.append(t_us(g1))
    a, b = sorted(a)[4], sorted(b)[4]
    print(f"{M} x {N} x {K}: equal {eq}   row-major scratch {a:8.2f} us   tile-major scratch {b:8.2f} us   ({100 * (a / b - 1):+.1f} %)", flush=True)
    del x, w, wt, o0, o1, g0, g1
    torch.cuda.empty_cache()
