#!/usr/bin/env python3
"""Where does "decode once + dense GEMM" beat the fused kernels?  One GPU, one process, interleaved rounds.

Legs per shape (us per call, median of the rounds):
  fused      mbnb_matmul_4bit_ex with MBNB_MATMUL_FUSED_ONLY: the fused / split-K kernels
  dense s=S  mbnb_dequantize_4bit + mbnb_gemm_dense with S split-K slices
  auto       bnb.matmul_4bit (what the library picks)
  blas       mbnb_dequantize_4bit + torch.matmul (vendor BLAS)

    python tools/exp/sweep_dense.py [--shapes "M,N,K;..."] [--rounds 5] [--iters 20]
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
from mps_bitsandbytes_amd import _native  # noqa: E402
from mps_bitsandbytes_amd import functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="256,4096,4096;512,4096,4096;768,4096,4096;1024,4096,4096;1536,4096,4096;2048,4096,4096;"
                "3072,4096,4096;4096,4096,4096;8192,4096,4096;512,11008,4096;1024,11008,4096;2048,11008,4096;4096,11008,4096;"
                "1024,4096,11008;4096,4096,11008")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--slices", default="1,2,4,8")
ap.add_argument("--tiles", default="256,128", help="row extents of a tile to try")
args = ap.parse_args()
lib = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
BF16 = _native.DTYPE_CODE[torch.bfloat16]

for shp in args.shapes.split(";"):
    M, N, K = [int(v) for v in shp.split(",")]
    g = torch.Generator(device=dev)
    g.manual_seed(M + N)
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    packed, state = bnb.quantize_nf4(W, blocksize=64)
    X = torch.randn(M, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    Wd = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    keep = []
    desc = F._absmax_desc(state.absmax, state.state2, keep)
    ws_bytes = max(int(lib.mbnb_matmul_4bit_workspace_bytes(M, N, K, K, BF16, 0)), 16 * M * N * 4) + 256
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    assert ws.data_ptr() % 256 == 0

    def fused():
        rc = lib.mbnb_matmul_4bit(X.data_ptr(), M, K, packed.data_ptr(), ctypes.byref(desc), N, K, 64, _native.QUANT_CODE["nf4"], BF16,
                                     None, BF16, out.data_ptr(), ws.data_ptr(), ws_bytes, 1, st)
        assert rc == 0, rc

    def dequant():
        rc = lib.mbnb_dequantize_4bit(packed.data_ptr(), ctypes.byref(desc), N, K, K, 64, _native.QUANT_CODE["nf4"], BF16, Wd.data_ptr(), st)
        assert rc == 0, rc

    def dense(s, tile_m=256):
        dequant()
        rc = lib.mbnb_gemm_dense(X.data_ptr(), Wd.data_ptr(), BF16, None, BF16, out.data_ptr(), M, N, K, K, ws.data_ptr(), ws_bytes,
                                 s | ((tile_m // 128) << 8), st)
        assert rc == 0, rc

    fused()
    torch.cuda.synchronize()
    ref = out.clone()
    name_fused = _native.last_kernel()
    legs = {"fused": fused, "auto": lambda: bnb.matmul_4bit(X, packed, state), "blas": lambda: (dequant(), torch.matmul(X, Wd.t(), out=out))}
    checks = []
    for tile_m in [int(v) for v in args.tiles.split(",")]:
        for s in [int(v) for v in args.slices.split(",")]:
            if s > 1 and s * 512 > K:
                continue
            out.fill_(float("nan"))
            dense(s, tile_m)
            torch.cuda.synchronize()
            rel = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
            tag = f"t{tile_m} s={s}"
            checks.append(f"{tag}: {'bit-equal' if torch.equal(out, ref) else f'rel {rel:.1e}'}")
            legs[f"dense {tag}"] = (lambda s=s, tile_m=tile_m: dense(s, tile_m))
    y = bnb.matmul_4bit(X, packed, state)
    torch.cuda.synchronize()
    auto_name = _native.last_kernel()
    for f in legs.values():
        for _ in range(10):
            f()
    torch.cuda.synchronize()
    times = {k: [] for k in legs}
    for r in range(args.rounds):
        for k, f in legs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                f()
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            e1.synchronize()
            times[k].append(e0.elapsed_time(e1) / args.iters * 1e3)
    med = {k: statistics.median(t) for k, t in times.items()}
    best = min((k for k in med if k.startswith("dense")), key=lambda k: med[k])
    line = f"{M:6d} x {N:6d} x {K:6d}  " + "  ".join(f"{k} {v:7.1f}" for k, v in med.items())
    line += f"   | fused = {name_fused}; best dense: {best} ({med[best] / med['fused']:.2f} of fused)  auto -> {auto_name}, "
    line += f"equal to fused: {torch.equal(y, ref)}; {', '.join(checks)}"
    print(line, flush=True)
