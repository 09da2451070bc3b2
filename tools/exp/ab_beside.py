"""A/B of the decode-beside path (k_decode_beside on a side stream + k_gemm_gated, csrc/gemm_beside.h; functional.DECODE_BESIDE)
against the two-launch form (dequantize_4bit + k_gemm_dense): bit equality (several shapes, with / without bias, f16 / bf16,
NF4 / FP4, plain / double-quantised absmax, different weights through the same scratch), flags back to zero, then interleaved
timing at 4096^3 bf16 and at BASELINE configs[2] (11008 x 4096, double-quantised)."""
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


def both(x, packed, st, bias=None, gemm_first=False, side=False, split=False):
    F.DECODE_BESIDE = True; F.BESIDE_GEMM_FIRST = gemm_first; F.BESIDE_SIDE_STREAM = side; F.BESIDE_SPLIT = split
    y1 = bnb.matmul_4bit(x, packed, st, bias); k1 = _native.last_kernel()
    F.DECODE_BESIDE = False; F.BESIDE_GEMM_FIRST = False; F.BESIDE_SIDE_STREAM = False; F.BESIDE_SPLIT = False
    y0 = bnb.matmul_4bit(x, packed, st, bias); k0 = _native.last_kernel()
    torch.cuda.synchronize()
    return y0, y1, k0, k1


ok = True
for (M, N, K, dt, qt, wb, dq) in [(4096, 4096, 4096, torch.bfloat16, "nf4", False, False), (4000, 4096, 2048, torch.float16, "fp4", True, False),
                                  (3900, 2560, 4096, torch.bfloat16, "nf4", True, True), (4096, 1000, 2048, torch.bfloat16, "nf4", False, False),
                                  (2048, 4096, 4096, torch.bfloat16, "nf4", False, False), (1536, 11008, 4096, torch.bfloat16, "nf4", False, True),
                                  (4096, 11008, 4096, torch.bfloat16, "nf4", False, True), (8192, 4096, 1024, torch.float16, "nf4", True, False)]:
    for rep in range(4):
        g = torch.Generator(device=dev); g.manual_seed(M + N + K + rep)
        W = torch.randn(N, K, generator=g, device=dev).to(dt) * (0.05 if dq else 1.0)
        x = torch.randn(M, K, generator=g, device=dev).to(dt)
        bias = torch.randn(N, generator=g, device=dev).to(dt) if wb else None
        packed, st = bnb.quantize_4bit(W, blocksize=64, quant_type=qt, compress_statistics=dq)
        y0, y1, k0, k1 = both(x, packed, st, bias, gemm_first=(rep == 2), side=(rep == 1), split=(rep == 3))
        eq = torch.equal(y0, y1)
        ok &= eq
        print(M, N, K, dt, qt, "bias" if wb else "", "dq" if dq else "", k0, k1, "equal:", eq, "sync words set:", F.in_launch_errors(), flush=True)
        if not eq:
            d = (y0.float() - y1.float()).abs()
            bad = (d > 0).nonzero()
            print("  mismatches:", bad.shape[0], "cols:", torch.unique(bad[:, 1])[:12].tolist(), "rows:", torch.unique(bad[:, 0])[:12].tolist(), flush=True)
print("all equal:", ok, flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "time":
    for (M, N, K, dq) in [(4096, 4096, 4096, False), (4096, 11008, 4096, True), (2048, 4096, 4096, False)]:
        g = torch.Generator(device=dev); g.manual_seed(1)
        W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16) * (0.05 if dq else 1.0)
        x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        packed, st = bnb.quantize_nf4(W, blocksize=64, compress_statistics=dq)

        def leg(beside, gemm_first=False, side=False, split=False):
            def f():
                F.DECODE_BESIDE = beside; F.BESIDE_GEMM_FIRST = gemm_first; F.BESIDE_SIDE_STREAM = side; F.BESIDE_SPLIT = split
                bnb.matmul_4bit(x, packed, st)
            return f
        legs = {"two launches": leg(False), "decode beside (one stream, any-order)": leg(True), "decode beside (two streams)": leg(True, False, True),
                "decode beside (two streams, GEMM first)": leg(True, True), "decode beside (split)": leg(True, False, False, True)}
        for f in legs.values():
            for _ in range(30):
                f()
        ev(legs["two launches"], 2000)
        res = {k: [] for k in legs}
        for rep in range(7):
            for k, f in legs.items():
                res[k].append(ev(f, 200))
        for k, v in res.items():
            v = sorted(v)
            print(f"{M} x {N} x {K}{' dq' if dq else ''}: {k:44s} median {v[3]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)
    F.DECODE_BESIDE = False; F.BESIDE_GEMM_FIRST = False; F.BESIDE_SPLIT = False
    print("sync words set:", F.in_launch_errors(), flush=True)
