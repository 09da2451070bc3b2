"""A/B of k_gemm_dense128 (128 x 128 tiles, three stages; csrc/gemm_dense128.h) against k_gemm_dense with the library's plan (256-wide
tiles, K slices where it splits) on an already dequantised weight: bit equality with the UNSPLIT 256 x 256 result, then timing
(HIP graph of 20 calls: device time per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = _native.lib()
dt = torch.bfloat16
shapes = [(512, 4096, 4096), (768, 4096, 4096), (1024, 4096, 4096), (1536, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096),
          (1024, 11008, 4096), (1000, 2600, 1024), (515, 1000, 640)]


def run(x, w, bias, out, M, N, K, ldw, ws, slices, tile):
    rc = lib.mbnb_gemm_dense(x.data_ptr(), w.data_ptr(), 1, None if bias is None else bias.data_ptr(), 1, out.data_ptr(), M, N, K, ldw,
                             None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), slices | (tile << 8), _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())


def graph_time(fn):
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


for (M, N, K) in shapes:
    gen = torch.Generator(device=dev); gen.manual_seed(M + N)
    x = torch.randn(M, K, generator=gen, device=dev).to(dt)
    w = (torch.randn(N, K, generator=gen, device=dev) * 0.05).to(dt)
    bias = torch.randn(N, generator=gen, device=dev).to(dt)
    ws_bytes = int(lib.mbnb_gemm_dense_workspace_bytes(M, N, K))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    o_ref = torch.empty(M, N, dtype=dt, device=dev); o_128 = torch.full((M, N), float("nan"), dtype=dt, device=dev); o_plan = torch.empty(M, N, dtype=dt, device=dev)
    run(x, w, bias, o_ref, M, N, K, K, None, 1, 2)        # unsplit 256 x 256
    run(x, w, bias, o_128, M, N, K, K, None, 1, 3)        # 128 x 128
    torch.cuda.synchronize()
    eq = torch.equal(o_ref, o_128)
    t_plan = graph_time(lambda: run(x, w, None, o_plan, M, N, K, K, ws, 0, 0))
    t_128 = graph_time(lambda: run(x, w, None, o_128, M, N, K, K, None, 1, 3))
    t_256 = graph_time(lambda: run(x, w, None, o_ref, M, N, K, K, None, 1, 2))
    print(f"{M:5d} x {N:5d} x {K:5d}: 128-tile == 256-tile: {eq}   plan {t_plan:7.2f} us   256x256 unsplit {t_256:7.2f}   128x128 {t_128:7.2f}", flush=True)
