"""Linear8bit forward time vs M at 4096x4096 (which kernel serves which batch size)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
lin = torch.nn.Linear(4096, 4096, bias=False).to(torch.bfloat16).to(dev)
l8 = bnb.Linear8bit.from_linear(lin)
for M in (1, 2, 4, 8, 16, 32, 64, 128, 512, 2048, 4096):
    x = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        l8(x)
        with torch.cuda.graph(g, stream=s):
            for _ in range(8): l8(x)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 / 8 * 1e3
    print(f"M={M}: {us:.2f} us ({_native.last_kernel()})", flush=True)
