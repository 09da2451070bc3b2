"""Wall time per step of the headline GEMM: eager launches vs one HIP graph of all steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import synthetic
dev = "cuda"
W = synthetic.normal((4096, 4096), torch.bfloat16, 1, 0.02).to(dev)
X = synthetic.normal((4096, 4096), torch.bfloat16, 2, 1.0).to(dev)
packed, st = bnb.quantize_nf4(W)
def step(): return bnb.matmul_4bit(X, packed, st)
for steps in (10, 50, 200):
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e6
    t0 = time.perf_counter()
    for _ in range(steps): step()
    cpu_only = (time.perf_counter() - t0) / steps * 1e6
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        step()
        with torch.cuda.graph(g, stream=side):
            for _ in range(steps): step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / steps * 1e6
    print(f"steps={steps}: eager {eager:.1f} us/step (cpu enqueue {cpu_only:.1f}), graph {graph:.1f} us/step", flush=True)
