#!/usr/bin/env python3
"""Host-side cost of one eager matmul_4bit / Linear4bit call at M = 1 (the kernel itself takes ~5 us): wall time per call of
a Python loop, and a cProfile breakdown of the wrapper."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
W = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
packed, st = bnb.quantize_nf4(W)
x = torch.randn(1, 4096, device=dev, dtype=torch.bfloat16)
lin = bnb.Linear4bit.from_linear(torch.nn.Linear(4096, 4096, bias=False).to(torch.bfloat16).to(dev))
for name, fn in (("matmul_4bit", lambda: bnb.matmul_4bit(x, packed, st)), ("Linear4bit.forward", lambda: lin(x)),
                 ("torch F.linear bf16 (context)", lambda: torch.nn.functional.linear(x, W))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per eager call")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    bnb.matmul_4bit(x, packed, st)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
