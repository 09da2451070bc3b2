"""Per-kernel floor inside a HIP graph (64 back-to-back tiny launches), next to the M = 1 GEMV."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
dev = "cuda"
def graph_time(fn, n=64):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 / n * 1e3
t = torch.zeros(64, device=dev)
print(f"tiny elementwise kernel (64 floats): {graph_time(lambda: t.add_(1.0)):.2f} us per launch")
W = torch.randn(64, 128, device=dev).to(torch.bfloat16); p, st = bnb.quantize_nf4(W); x = torch.randn(1, 128, device=dev).to(torch.bfloat16)
print(f"gemv on a 64 x 128 weight: {graph_time(lambda: bnb.matmul_4bit(x, p, st)):.2f} us per launch")
W = torch.randn(4096, 4096, device=dev).to(torch.bfloat16); p, st = bnb.quantize_nf4(W); x = torch.randn(1, 4096, device=dev).to(torch.bfloat16)
print(f"gemv 4096 x 4096 (single hot layer, LLC-resident): {graph_time(lambda: bnb.matmul_4bit(x, p, st)):.2f} us per launch")
big = torch.empty(9453568 // 4, device=dev); out = torch.empty_like(big)
print(f"device copy of 9.45 MB (read + write): {graph_time(lambda: out.copy_(big)):.2f} us per launch")
print(f"sum over 9.45 MB: {graph_time(lambda: big.sum()):.2f} us per launch (two kernels)")
