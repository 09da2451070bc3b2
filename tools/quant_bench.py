"""quantize_4bit / dequantize_4bit / quantize_rowwise / double-quant kernel times at 4096 x 4096 (HBM-bound byte work)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
dev = "cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.float16, torch.bfloat16):
    W = torch.randn(4096, 4096, device=dev).to(dt)
    p, st = bnb.quantize_nf4(W)
    pc, stc = bnb.quantize_nf4(W, compress_statistics=True)
    algo = 4096 * 4096 * 2 + 4096 * 4096 // 2 + 4096 * 64 * 4
    for name, fn in (("quantize_nf4", lambda: bnb.quantize_nf4(W)), ("quantize_nf4 + double quant", lambda: bnb.quantize_nf4(W, compress_statistics=True)),
                     ("dequantize_nf4", lambda: bnb.dequantize_4bit(p, st)), ("dequantize_nf4 (double-quantised absmax)", lambda: bnb.dequantize_4bit(pc, stc)),
                     ("quantize_rowwise", lambda: bnb.quantize_rowwise(W)), ("quantize_fp8_e4m3", lambda: bnb.quantize_fp8_e4m3(W))):
        us = t(fn)
        print(f"{dt} {name}: {us:.1f} us  ({algo / us / 1e3:.0f} GB/s of 42.99 MB)", flush=True)
