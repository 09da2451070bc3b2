"""A/B of the four-wave fused kernel (k_gemm_fused4; the no-scratch kernel of mbnb_matmul_4bit since round 4) against the decode-once path (dequantize_4bit +
k_gemm_dense) and the 8-wave fused k_gemm256s (FUSED_ONLY): bit equality of the outputs, then interleaved timing (HIP events
around N launches, variants alternating inside one process).   python tools/exp/ab_fused4.py [check|time] [M N K] [dq]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, functional as F

dev = torch.device("cuda:0")
lib = _native.lib()


def call(x, packed, state, flags, bias=None, ws=None):
    N, K = state.shape
    M = x.shape[0]
    Kw = F._padded(K, state.blocksize)
    keep = []
    desc = F._absmax_desc(state.absmax, state.state2, keep)
    out = torch.empty(M, N, dtype=x.dtype, device=dev)
    code = _native.DTYPE_CODE[x.dtype]
    rc = lib.mbnb_matmul_4bit(x.data_ptr(), M, K, packed.data_ptr(), ctypes.byref(desc), N, Kw, state.blocksize,
                                 _native.QUANT_CODE[state.quant_type], code, None if bias is None else bias.data_ptr(), code,
                                 out.data_ptr(), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), flags,
                                 _native.stream_ptr(dev))
    assert rc == 0, (rc, lib.mbnb_last_error())
    return out


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (4096, 4096, 4096)
    dq = "dq" in sys.argv
    dt = torch.bfloat16
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    W = (torch.randn(N, K, generator=g, device=dev) * (0.02 if dq else 1.0)).to(dt)
    x = torch.randn(M, K, generator=g, device=dev).to(dt)
    bias = torch.randn(N, generator=g, device=dev).to(dt)
    packed, st = bnb.quantize_nf4(W, blocksize=64, compress_statistics=dq)
    Kw = F._padded(K, 64)
    ws = torch.empty(int(lib.mbnb_matmul_4bit_workspace_bytes(M, N, K, Kw, _native.DTYPE_CODE[dt], 0)), dtype=torch.uint8, device=dev)
    y_ref = call(x, packed, st, 0, None, ws)
    k_ref = _native.last_kernel()
    y_f4 = call(x, packed, st, 0, None, None)   # no workspace: the fused kernel (k_gemm_fused4 since round 4)
    k_f4 = _native.last_kernel()
    torch.cuda.synchronize()
    print("kernels:", k_ref, k_f4, flush=True)
    eq = torch.equal(y_ref, y_f4)
    print("equal (no bias):", eq, "max|d|:", (y_ref.float() - y_f4.float()).abs().max().item(), flush=True)
    if not eq:
        d = (y_ref.float() - y_f4.float()).abs()
        bad = (d > 0).nonzero()
        print("mismatches:", bad.shape[0], "first:", bad[:8].tolist(), flush=True)
        cols = torch.unique(bad[:, 1])
        rows = torch.unique(bad[:, 0])
        print("bad cols:", cols.numel(), cols[:16].tolist(), "bad rows:", rows.numel(), rows[:16].tolist())
    yb_ref = call(x, packed, st, 0, bias, ws)
    yb_f4 = call(x, packed, st, 0, bias, None)
    print("equal (bias):", torch.equal(yb_ref, yb_f4), flush=True)
    if mode != "time":
        return
    variants = {"dequant+dense": lambda: call(x, packed, st, 0, None, ws), "fused4": lambda: call(x, packed, st, 0, None, None),
                "fused 8-wave": lambda: call(x, packed, st, 1, None, None)}
    for f in variants.values():
        for _ in range(50):
            f()
    torch.cuda.synchronize()
    # settle the clock under load
    t0 = ev(variants["fused4"], 2000)
    res = {k: [] for k in variants}
    for rep in range(7):
        for k, f in variants.items():
            res[k].append(ev(f, 200))
    for k, v in res.items():
        v = sorted(v)
        print(f"{k:16s} median {v[len(v) // 2]:7.2f} us   min {v[0]:7.2f}   max {v[-1]:7.2f}")


main()
