#!/usr/bin/env python3
"""GEMV decode flavours (tools/exp/gemv_exp.hip): equality with the library result, then us per layer of a HIP graph over 64
distinct layers (HBM) and over one hot layer, interleaved."""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mps_bitsandbytes_amd as bnb
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemv_exp.so"))
lib.exp_gemv.restype = ctypes.c_int
lib.exp_gemv.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(0)
layers = []
for i in range(64):
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    p, st = bnb.quantize_nf4(W)
    layers.append((p, st.absmax.contiguous(), st))
x = torch.randn(1, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
out = torch.empty(1, N, dtype=torch.bfloat16, device=dev)
ref = bnb.matmul_4bit(x, layers[0][0], layers[0][2])
variants = [(0, 1), (10, 1), (0, 2)]     # (10, 1) = k_gemv4_lean
graphs = {}
for (dec, nr) in variants:
    s = torch.cuda.current_stream().cuda_stream
    lib.exp_gemv(dec, nr, x.data_ptr(), layers[0][0].data_ptr(), layers[0][1].data_ptr(), out.data_ptr(), N, K, s)
    torch.cuda.synchronize()
    print(f"dec {dec} nr {nr}: equal to the library result: {torch.equal(out, ref)}  max|d| {(out.float() - ref.float()).abs().max().item():.4g}  rel {((out.float() - ref.float()).norm() / ref.float().norm()).item():.3g}", flush=True)
    for hot in (False, True):
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                ss = torch.cuda.current_stream().cuda_stream
                for i in range(64):
                    p, a, _ = layers[0 if hot else i]
                    lib.exp_gemv(dec, nr, x.data_ptr(), p.data_ptr(), a.data_ptr(), out.data_ptr(), N, K, ss)
        torch.cuda.current_stream().wait_stream(side)
        graphs[(dec, nr, hot)] = gr
times = {k: [] for k in graphs}
for k, gr in graphs.items():
    for _ in range(5):
        gr.replay()
torch.cuda.synchronize()
for r in range(7):
    for k, gr in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); e1.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10 / 64 * 1e3)
for k in graphs:
    print(f"dec {k[0]} rows/wave {k[1]} {'hot layer ' if k[2] else '64 layers '}: median {statistics.median(times[k]):.3f} us per layer  min {min(times[k]):.3f}", flush=True)
