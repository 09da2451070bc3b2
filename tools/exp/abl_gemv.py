#!/usr/bin/env python3
"""Timing-only ablation of k_gemv4_lean (GV_ABL builds of csrc/gemv4_lean.h: libgemv_abl<bits>.so): us per layer of one HIP graph over 64 rotating
layers and over one hot layer, all variants alternating.  bits: 1 no lookups / products, 2 no activation reads, 4 no decode loop, 8 no weight loads."""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mps_bitsandbytes_amd as bnb
here = os.path.dirname(os.path.abspath(__file__))
variants = [0, 1, 2, 3, 4, 12]
libs = {}
for v in variants:
    l = ctypes.CDLL(os.path.join(here, f"libgemv_abl{v}.so"))
    l.exp_gemv.restype = ctypes.c_int
    l.exp_gemv.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
    libs[v] = l
dev = torch.device("cuda:0")
N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(0)
layers = []
for i in range(64):
    W = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    p, st = bnb.quantize_nf4(W)
    layers.append((p, st.absmax.contiguous()))
x = torch.randn(1, K, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
out = torch.empty(1, N, dtype=torch.bfloat16, device=dev)
graphs = {}
for v in variants:
    for hot in (False, True):
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            libs[v].exp_gemv(10, 1, x.data_ptr(), layers[0][0].data_ptr(), layers[0][1].data_ptr(), out.data_ptr(), N, K, torch.cuda.current_stream().cuda_stream)
            with torch.cuda.graph(gr, stream=side):
                ss = torch.cuda.current_stream().cuda_stream
                for i in range(64):
                    p, a = layers[0 if hot else i]
                    libs[v].exp_gemv(10, 1, x.data_ptr(), p.data_ptr(), a.data_ptr(), out.data_ptr(), N, K, ss)
        torch.cuda.current_stream().wait_stream(side)
        graphs[(v, hot)] = gr
for gr in graphs.values():
    for _ in range(10):
        gr.replay()
torch.cuda.synchronize()
times = {k: [] for k in graphs}
for r in range(7):
    for k, gr in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); e1.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10 / 64 * 1e3)
# residency experiment (full kernel, GV_ABL 0): dynamic LDS per workgroup -> resident workgroups per CU (160 KiB / LDS): does a grid that arrives in
# several rounds overlap the next round's loads with this round's decode?
res_graphs = {}
for kib in (8, 40, 53, 80, 159):
    for hot in (False, True):
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rc = libs[0].exp_gemv(100 + kib, 1, x.data_ptr(), layers[0][0].data_ptr(), layers[0][1].data_ptr(), out.data_ptr(), N, K, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            with torch.cuda.graph(gr, stream=side):
                ss = torch.cuda.current_stream().cuda_stream
                for i in range(64):
                    p, a = layers[0 if hot else i]
                    libs[0].exp_gemv(100 + kib, 1, x.data_ptr(), p.data_ptr(), a.data_ptr(), out.data_ptr(), N, K, ss)
        torch.cuda.current_stream().wait_stream(side)
        res_graphs[(kib, hot)] = gr
for gr in res_graphs.values():
    for _ in range(10):
        gr.replay()
torch.cuda.synchronize()
rt = {k: [] for k in res_graphs}
for r in range(7):
    for k, gr in res_graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); e1.synchronize()
        rt[k].append(e0.elapsed_time(e1) / 10 / 64 * 1e3)
for kib in (8, 40, 53, 80, 159):
    print(f"full kernel, {kib:3d} KiB of LDS per workgroup ({min(4 if kib <= 40 else 160 // kib, 160 // kib)} resident per CU)   rotating {statistics.median(rt[(kib, False)]):.3f} us   hot {statistics.median(rt[(kib, True)]):.3f} us", flush=True)
names = {0: "full kernel", 1: "no lookups / products", 2: "no activation reads", 3: "neither (dot2 + extraction only)", 4: "no decode loop (loads, wait, reduce, store)",
         12: "no decode, no weight loads (x to LDS, table, barrier, reduce, store)"}
for v in variants:
    print(f"GV_ABL {v:2d}  {names[v]:70s}  rotating {statistics.median(times[(v, False)]):.3f} us   hot {statistics.median(times[(v, True)]):.3f} us", flush=True)
