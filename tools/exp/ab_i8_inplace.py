"""A/B of matmul_int8 at 4096^3: workspace given -> transpose + k_gemm_dense<I8>; none -> k_gemm_i8_inplace (four waves, B read
in place).  Bit equality between the two and against the exact integer formula on a row sample, then interleaved timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native

dev = torch.device("cuda:0")
lib = _native.lib()
shapes = [(4096, 4096, 4096), (2048, 3088, 384), (2500, 2608, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]


def call(A, B, sa, sb, odt, ws):
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty(M, N, dtype=odt, device=dev)
    rc = lib.mbnb_matmul_int8(A.data_ptr(), B.data_ptr(), sa.data_ptr(), sb.data_ptr(), M, N, K, _native.DTYPE_CODE[odt], out.data_ptr(),
                              None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())
    return out


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in shapes:
    g = torch.Generator(device=dev); g.manual_seed(M + N)
    A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
    B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
    sa = torch.rand(M, generator=g, device=dev) + 0.5
    sb = torch.rand(N, generator=g, device=dev) + 0.5
    ws = torch.empty(N * K, dtype=torch.uint8, device=dev)
    for odt in (torch.float16, torch.bfloat16, torch.float32):
        y_ws = call(A, B, sa, sb, odt, ws); k_ws = _native.last_kernel()
        y_ip = call(A, B, sa, sb, odt, None); k_ip = _native.last_kernel()
        torch.cuda.synchronize()
        rows = torch.arange(0, M, max(1, M // 48), device=dev)[:48]
        acc = (A[rows].double() @ B.double())
        ref = (acc.float() * (sa[rows] / 127.0)[:, None] * (sb / 127.0)[None, :]).to(odt)
        print(M, N, K, odt, k_ws, k_ip, "equal:", torch.equal(y_ws, y_ip), "vs exact:", torch.equal(y_ip[rows], ref), flush=True)
    if (M, N, K) == (4096, 4096, 4096):
        legs = {"transpose+dense": lambda: call(A, B, sa, sb, torch.float16, ws), "in place (4 waves)": lambda: call(A, B, sa, sb, torch.float16, None)}
        for f in legs.values():
            for _ in range(30):
                f()
        ev(legs["in place (4 waves)"], 3000)
        res = {k: [] for k in legs}
        for rep in range(7):
            for k, f in legs.items():
                res[k].append(ev(f, 200))
        for k, v in res.items():
            v = sorted(v)
            print(f"{k:20s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)
