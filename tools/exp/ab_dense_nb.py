"""A/B of the column-balanced grids of k_gemm_dense_nb (round 4; csrc/gemm_dense.h, plan: gemm_dense_nb_plan in gemm_dense.hip) against the
uniform 256 x 256 tiles of k_gemm_dense on an already dequantised weight: bit equality, then device time per call (HIP graph of 20 calls,
median of 5 replays, the two variants alternating).   python tools/exp/ab_dense_nb.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = _native.lib()
dt = torch.bfloat16
# LLM layer shapes (M rows x N outputs x K inputs): Llama-7B / 13B / 70B / Mistral FFN and attention projections, an LM head, odd row counts
shapes = [(4096, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 11008), (4096, 13824, 5120), (4096, 5120, 13824), (4096, 5120, 5120),
          (4096, 14336, 4096), (4096, 28672, 8192), (4096, 6144, 4096), (4096, 12288, 4096), (4096, 22016, 4096), (4096, 32000, 4096),
          (2048, 11008, 4096), (8192, 11008, 4096), (3000, 11008, 4096), (5000, 5120, 5120), (4096, 27648, 5120), (4096, 10240, 8192),
          (4096, 9216, 4096), (8192, 5120, 5120), (6144, 6144, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:6]


def run(x, w, bias, out, M, N, K, slices, tile):
    rc = lib.mbnb_gemm_dense(x.data_ptr(), w.data_ptr(), 1, None if bias is None else bias.data_ptr(), 1, out.data_ptr(), M, N, K, K,
                             None, 0, slices | (tile << 8), _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())


def graph(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    return g


def t_us(g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


print(f"{'M x N x K':>22s}  {'plan':>22s}  equal   uniform us   256x128 us   plan us   gain vs best other   TFLOP/s (plan)  frac", flush=True)
for (M, N, K) in shapes:
    gen = torch.Generator(device=dev); gen.manual_seed(M + N + K)
    x = torch.randn(M, K, generator=gen, device=dev).to(dt)
    w = (torch.randn(N, K, generator=gen, device=dev) * 0.05).to(dt)
    bias = torch.randn(N, generator=gen, device=dev).to(dt)
    o_u = torch.full((M, N), float("nan"), dtype=dt, device=dev)
    o_p = torch.full((M, N), float("nan"), dtype=dt, device=dev)
    run(x, w, bias, o_u, M, N, K, 1, 2)        # uniform 256 x 256
    run(x, w, bias, o_p, M, N, K, 1, 0)        # the library's choice at one slice
    plan = _native.last_kernel()
    torch.cuda.synchronize()
    eq = torch.equal(o_u, o_p) and bool(torch.isfinite(o_p.float()).all())
    gu, gh, gp = graph(lambda: run(x, w, None, o_u, M, N, K, 1, 2)), graph(lambda: run(x, w, None, o_u, M, N, K, 1, 1)), graph(lambda: run(x, w, None, o_p, M, N, K, 1, 0))
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:       # the clock needs load to settle
        gu.replay(); gh.replay(); gp.replay()
        torch.cuda.synchronize()
    tu, th, tp = [], [], []
    for _ in range(5):
        tu.append(t_us(gu)); th.append(t_us(gh)); tp.append(t_us(gp))
    tu, th, tp = sorted(tu)[2], sorted(th)[2], sorted(tp)[2]
    tf = 2.0 * M * N * K / (tp * 1e-6) / 1e12
    print(f"{M:6d} x {N:6d} x {K:6d}  {plan:>22s}  {str(eq):5s}  {tu:10.2f}  {th:10.2f}  {tp:8.2f}  {100 * (min(tu, th) / tp - 1):5.1f}%  {tf:10.1f}  {tf / 2500:.3f}", flush=True)
    del x, w, o_u, o_p, gu, gh, gp
    torch.cuda.empty_cache()
