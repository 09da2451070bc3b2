"""Per-k-step time of the 4096x4096 NF4 GEMM for row strides K that are / are not powers of two
(L2 channel hot-spotting probe).  Env switches (MBNB_ABLATE with an ablation build) are inherited."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
for K in (4096, 4160, 4224, 4352, 4608, 5120, 8192, 8256):
    W = (torch.randn(4096, K, device=dev) * 0.02).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W)
    X = torch.randn(4096, K, device=dev).to(torch.bfloat16)
    for _ in range(30): bnb.matmul_4bit(X, packed, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): bnb.matmul_4bit(X, packed, st)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    print(f"K={K}: {best:.1f} us  {best / (K / 64) * 1e3:.0f} ns/k-step  {_native.last_kernel()}", flush=True)
