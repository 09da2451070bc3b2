#!/usr/bin/env python3
"""A/B of the int8 256 x 256 kernels (tools/exp/i8_exp.hip): exact equality with the round-1 path (integer contraction,
same epilogue arithmetic) and interleaved timing."""
import ctypes, os, statistics, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libi8_exp.so"))
lib.exp_i8.restype = ctypes.c_int
lib.exp_i8.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
shapes = [(4096, 4096, 4096), (2560, 2816, 512), (2560, 2560, 640), (2500, 2608, 384), (4096, 11008, 4096)]
variants = [0, 1, 3, 6]
g = torch.Generator(device=dev); g.manual_seed(1)
for (M, N, K) in shapes:
    A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
    B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
    Bt = B.t().contiguous()
    sa = torch.rand(M, generator=g, device=dev) + 0.5
    sb = torch.rand(N, generator=g, device=dev) + 0.5
    ws = torch.empty(N * K, dtype=torch.int8, device=dev)
    outs = {}
    for v in variants:
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        rc = lib.exp_i8(v, A.data_ptr(), B.data_ptr(), Bt.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), ws.data_ptr(), M, N, K, st)
        assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        outs[v] = out
    ref = ((A[:64].double() @ B.double()) * (sa[:64].double() / 127)[:, None] * (sb.double() / 127)[None, :])
    e = ((outs[0][:64].double() - ref).norm() / ref.norm()).item()
    print(f"{M}x{N}x{K}: round-1 path vs f64 formula on 64 rows: rel {e:.2e};", " ".join(f"v{v}=={'OK' if torch.equal(outs[v], outs[0]) else 'DIFF(' + str(int((outs[v] != outs[0]).sum())) + ')'}" for v in variants[1:]), flush=True)
    if (M, N, K) in ((4096, 4096, 4096), (4096, 11008, 4096)):
        times = {v: [] for v in variants}
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        for v in variants:
            for _ in range(100):
                lib.exp_i8(v, A.data_ptr(), B.data_ptr(), Bt.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), ws.data_ptr(), M, N, K, st)
        for r in range(7):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    lib.exp_i8(v, A.data_ptr(), B.data_ptr(), Bt.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), ws.data_ptr(), M, N, K, st)
                e1.record(); e1.synchronize()
                times[v].append(e0.elapsed_time(e1) / 30 * 1e3)
        for v in variants:
            med = statistics.median(times[v])
            print(f"   variant {v}: median {med:7.2f} us  min {min(times[v]):7.2f}   {2.0 * M * N * K / med / 1e6:7.1f} TOP/s  frac {2.0 * M * N * K / med / 1e6 / 5000:.3f}", flush=True)
