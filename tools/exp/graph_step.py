"""Step time of the headline workload (dequantize_4bit into the scratch + k_gemm_dense) issued eagerly vs replayed from one
HIP graph of 20 steps: how much of the ~4.8 us per step between the kernels is the launch path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
M = N = K = 4096
W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16)
p, st = bnb.quantize_nf4(W, blocksize=64)
X = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


step = lambda: bnb.matmul_4bit(X, p, st)
for _ in range(2000):
    step()
torch.cuda.synchronize()
eager = sorted(ev(step, 20) for _ in range(7))[3]
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(20):
            step()
torch.cuda.current_stream().wait_stream(side)
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
gr = sorted(ev(graph.replay, 5) / 20 for _ in range(7))[3]
print("eager %.2f us/step, graph of 20 steps %.2f us/step (%s)" % (eager, gr, _native.last_kernel()))
