"""Per-width tile times of k_gemm_dense_nb on forced grids (diagnostic tile codes 5-7 of mbnb_gemm_dense: all columns 160 / 192 / 224 wide, or
`a` columns one step wider first) against uniform 256-wide columns: device time per call from a HIP graph of 20 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = _native.lib()
dt = torch.bfloat16


def run(x, w, out, M, N, K, code, a=0):
    rc = lib.mbnb_gemm_dense(x.data_ptr(), w.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, None, 0, 1 | (code << 8) | (a << 16), _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())


def graph_us(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[2]


cases = [(4096, 4096, 4096, [(2, 0)]), (4096, 3584, 4096, [(2, 0), (7, 0)]), (4096, 3072, 4096, [(2, 0), (6, 0)]), (4096, 2560, 4096, [(2, 0), (5, 0)]),
         (4096, 3840, 4096, [(2, 0), (7, 8)]), (4096, 7168, 4096, [(2, 0), (7, 0)]), (4096, 7680, 4096, [(2, 0), (7, 16)]),
         (4096, 10752, 4096, [(2, 0), (7, 0)]), (4096, 12288, 4096, [(2, 0)]), (4096, 11008, 4096, [(2, 0), (7, 8), (7, 15), (7, 0)]),
         (4096, 22016, 4096, [(2, 0), (7, 16), (7, 0)]), (4096, 14336, 4096, [(2, 0), (7, 0), (7, 32)])]
for (M, N, K, variants) in cases:
    gen = torch.Generator(device=dev); gen.manual_seed(N)
    x = torch.randn(M, K, generator=gen, device=dev).to(dt)
    w = (torch.randn(N, K, generator=gen, device=dev) * 0.05).to(dt)
    ref = None
    line = f"{M} x {N} x {K}:"
    for code, a in variants:
        out = torch.full((M, N), float("nan"), dtype=dt, device=dev)
        run(x, w, out, M, N, K, code, a)
        name = _native.last_kernel()
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            eq = True
        else:
            eq = torch.equal(ref, out)
        t = graph_us(lambda: run(x, w, out, M, N, K, code, a))
        line += f"   [{name}] {t:.1f} us ({2.0 * M * N * K / t / 1e6:.0f} TF/s){'' if eq else ' MISMATCH'}"
    print(line, flush=True)
    del x, w, ref, out
    torch.cuda.empty_cache()
