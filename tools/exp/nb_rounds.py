"""Cost of a ROUND of tiles by tile width: M = 4096, K = 4096, N = R rounds x 16 columns -- all 256 wide, all 224 wide, all 192 wide, and one round
of 256-wide tiles followed by R - 1 rounds of 224-wide ones (forced grids: diagnostic tile codes of mbnb_gemm_dense).  Each figure: HIP graph of 10 calls,
200 ms of replays first (the clock needs load to settle), median of 7 replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = _native.lib()
dt = torch.bfloat16
M = K = 4096


def run(x, w, out, N, code, a=0):
    rc = lib.mbnb_gemm_dense(x.data_ptr(), w.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, None, 0, 1 | (code << 8) | (a << 16), _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())


def measure(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        g.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[3]


gen = torch.Generator(device=dev); gen.manual_seed(7)
x = torch.randn(M, K, generator=gen, device=dev).to(dt)
wmax = (torch.randn(7 * 4096, K, generator=gen, device=dev) * 0.05).to(dt)
print("rounds   all-256   per round | all-224   per round | all-192   per round | 256 then 224s   uniform-256 on that N", flush=True)
for R in range(1, 8):
    row = f"{R:6d}"
    for width, code in ((256, 2), (224, 7), (192, 6)):
        N = 16 * width * R
        out = torch.empty(M, N, dtype=dt, device=dev)
        t = measure(lambda: run(x, wmax[:N], out, N, code, 0))
        row += f"   {t:7.1f}   {t / R:7.1f}  |"
    N = 4096 + 3584 * (R - 1)
    out = torch.empty(M, N, dtype=dt, device=dev)
    tm = measure(lambda: run(x, wmax[:N], out, N, 7, 16))
    name = _native.last_kernel()
    tu = measure(lambda: run(x, wmax[:N], out, N, 2, 0))
    row += f"   {tm:7.1f} [{name}]   {tu:7.1f}"
    print(row, flush=True)
