"""Device time of the quantisers at 4096^2 (HIP graph of 20 calls: no host time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def graphed(fn):
    gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    return sorted(ev(gr.replay, 5) for _ in range(5))[2] / 20
for dt in (torch.float16, torch.bfloat16):
    W = torch.randn(4096, 4096, device=dev).to(dt)
    for name, fn in (("quantize_nf4", lambda: bnb.quantize_nf4(W)), ("quantize_nf4 + double quant", lambda: bnb.quantize_nf4(W, compress_statistics=True)),
                     ("quantize_rowwise", lambda: bnb.quantize_rowwise(W)), ("quantize_fp8_e4m3", lambda: bnb.quantize_fp8_e4m3(W))):
        print("%s %s: %.2f us" % (dt, name, graphed(fn)), flush=True)
for dt in (torch.float16,):
    W = torch.randn(4096, 4096, device=dev).to(dt)
    print("%s double_quant (row + column statistics, both int8 copies): %.2f us" % (dt, graphed(lambda: bnb.double_quant(W))), flush=True)
