"""Quick parity check of the skinny kernel against the oracle for several (M, N, K) and both 16-bit types."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic
def rel(y, r): return ((y.double() - r.double()).norm() / r.double().norm()).item()
for (M, N, K, dt, cs, qt, bias) in [(2, 64, 128, torch.float16, False, "nf4", False), (5, 512, 1024, torch.bfloat16, False, "nf4", True),
                                     (16, 4096, 4096, torch.bfloat16, False, "nf4", False), (17, 300, 256, torch.float16, True, "fp4", True),
                                     (33, 1000, 512, torch.bfloat16, True, "nf4", False), (64, 4096, 4096, torch.float16, False, "nf4", True)]:
    W = synthetic.normal((N, K), dt, seed=3, std=0.05)
    x = synthetic.normal((M, K), dt, seed=4)
    b = synthetic.normal((N,), dt, seed=5) if bias else None
    packed, st = bnb.quantize_4bit(W.cuda(), blocksize=64, quant_type=qt, compress_statistics=cs)
    y = bnb.matmul_4bit(x.cuda(), packed, st, None if b is None else b.cuda()).cpu()
    kern = _native.last_kernel()
    deq = bnb.dequantize_4bit(packed, st).cpu()
    ref = (x.float() @ deq.float().t() + (b.float() if bias else 0)).to(dt)
    print(M, N, K, dt, cs, qt, kern, "rel", rel(y, ref), flush=True)
