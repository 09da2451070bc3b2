#!/usr/bin/env python3
"""A/B of the 256 x 256 fused NF4 GEMM variants (tools/exp/gemm_exp.hip) on one GPU, one process:
bit-equality of every variant against the shipping k_gemm256p, then interleaved timing rounds (median / min us).

    python tools/exp/ab_gemm.py [--variants -1,0,1,2,3] [--rounds 7] [--iters 40] [--shape M,N,K]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mps_bitsandbytes_amd as bnb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="-1,0,1,2,3,4,5,6,7")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--shape", default="4096,4096,4096")
ap.add_argument("--check-shapes", default="2560,2560,512;2500,2600,256;4096,4096,4096;512,11008,4096")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_exp.so"))
lib.exp_gemm256.restype = ctypes.c_int
lib.exp_gemm256.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def setup(M, N, K, seed=0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    packed, state = bnb.quantize_nf4(W, blocksize=64)
    X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    return X, packed, state.absmax.contiguous()


def run(v, X, packed, absmax, out, M, N, K):
    rc = lib.exp_gemm256(v, X.data_ptr(), packed.data_ptr(), absmax.data_ptr(), out.data_ptr(), M, N, K, K, st)
    assert rc == 0, f"variant {v}: rc {rc}"


# ---- correctness: bit-equal to the shipping kernel (same B-operand bits, same MFMA order per accumulator)
ok = True
for shp in args.check_shapes.split(";"):
    M, N, K = [int(v) for v in shp.split(",")]
    if ((M + 255) // 256) * ((N + 255) // 256) < 1 or K % 256:
        continue
    X, packed, absmax = setup(M, N, K, seed=M + N)
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run(-1, X, packed, absmax, ref, M, N, K)
    torch.cuda.synchronize()
    yref = bnb.matmul_4bit(X, packed, bnb.functional.QuantState(absmax=absmax, shape=torch.Size([N, K]), blocksize=64, quant_type="nf4", dtype=torch.bfloat16))
    for v in variants:
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        run(v, X, packed, absmax, out, M, N, K)
        torch.cuda.synchronize()
        same = torch.equal(out.view(torch.int16), ref.view(torch.int16))
        nbad = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
        rel = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
        print(f"check {shp} variant {v:2d}: {'bit-equal' if same else f'DIFFERS in {nbad} elements (rel {rel:.2e})'}", flush=True)
        ok = ok and (same or (v >= 100 and rel < 5e-5))
    print(f"check {shp}: harness k_gemm256p == library matmul_4bit: {torch.equal(ref, yref)}", flush=True)

# ---- timing: interleaved rounds
M, N, K = [int(v) for v in args.shape.split(",")]
X, packed, absmax = setup(M, N, K, seed=1)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for v in variants:           # warm-up + clock settle
    for _ in range(200):
        run(v, X, packed, absmax, out, M, N, K)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(args.rounds):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            run(v, X, packed, absmax, out, M, N, K)
        e0.record()
        for _ in range(args.iters):
            run(v, X, packed, absmax, out, M, N, K)
        e1.record()
        e1.synchronize()
        times[v].append(e0.elapsed_time(e1) / args.iters * 1e3)
flops = 2.0 * M * N * K
print(f"shape {M}x{N}x{K}, {args.rounds} interleaved rounds x {args.iters} launches (us per launch, incl. launch boundary)")
for v in variants:
    t = times[v]
    med = statistics.median(t)
    print(f"variant {v:2d}: median {med:7.2f}  min {min(t):7.2f}  max {max(t):7.2f}   {flops / med / 1e6:7.1f} TFLOP/s  frac {flops / med / 1e6 / 2500:.3f}", flush=True)
print("ALL BIT-EQUAL" if ok else "SOME VARIANT DIFFERS")
