"""A/B of the in-launch decode-once kernel (k_gemm_dq, csrc/gemm_dq.h; functional.DECODE_IN_LAUNCH) against the two-launch form
(dequantize_4bit + k_gemm_dense): bit equality (several seeds, with / without bias, f16 / bf16, NF4 / FP4), flags back to zero,
then interleaved timing at 4096^3 bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, functional as F

dev = torch.device("cuda:0")


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both(x, packed, st, bias=None):
    F.DECODE_IN_LAUNCH = True
    y1 = bnb.matmul_4bit(x, packed, st, bias); k1 = _native.last_kernel()
    F.DECODE_IN_LAUNCH = False
    y0 = bnb.matmul_4bit(x, packed, st, bias); k0 = _native.last_kernel()
    torch.cuda.synchronize()
    return y0, y1, k0, k1


ok = True
for (M, N, K, dt, qt, wb) in [(4096, 4096, 4096, torch.bfloat16, "nf4", False), (4000, 4096, 2048, torch.float16, "fp4", True),
                              (3900, 2560, 4096, torch.bfloat16, "nf4", True), (4096, 1000, 2048, torch.bfloat16, "nf4", False)]:
    g = torch.Generator(device=dev); g.manual_seed(M + N + K)
    W = torch.randn(N, K, generator=g, device=dev).to(dt)
    x = torch.randn(M, K, generator=g, device=dev).to(dt)
    bias = torch.randn(N, generator=g, device=dev).to(dt) if wb else None
    packed, st = bnb.quantize_4bit(W, blocksize=64, quant_type=qt)
    for rep in range(3):
        y0, y1, k0, k1 = both(x, packed, st, bias)
        eq = torch.equal(y0, y1)
        ok &= eq
        print(M, N, K, dt, qt, "bias" if wb else "", k0, k1, "equal:", eq, "sync words set:", F.in_launch_errors(), flush=True)
        if not eq:
            d = (y0.float() - y1.float()).abs()
            bad = (d > 0).nonzero()
            print("  mismatches:", bad.shape[0], "cols:", torch.unique(bad[:, 1])[:12].tolist(), "rows:", torch.unique(bad[:, 0])[:12].tolist(), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "time":
    M = N = K = 4096
    g = torch.Generator(device=dev); g.manual_seed(1)
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)

    def leg(flag):
        def f():
            F.DECODE_IN_LAUNCH = flag
            bnb.matmul_4bit(x, packed, st)
        return f
    legs = {"two launches": leg(False), "in-launch decode": leg(True)}
    for f in legs.values():
        for _ in range(30):
            f()
    ev(legs["two launches"], 3000)
    res = {k: [] for k in legs}
    for rep in range(7):
        for k, f in legs.items():
            res[k].append(ev(f, 200))
    for k, v in res.items():
        v = sorted(v)
        print(f"{k:18s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)
    print("sync words set after timing:", F.in_launch_errors())
