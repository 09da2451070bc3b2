"""Device time of dequantize_4bit at 4096^2 (HIP graph of 20 calls into a fixed output: no host time, no allocation)."""
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
for dt in (torch.float16, torch.bfloat16, torch.float32):
    W = torch.randn(4096, 4096, device=dev).to(dt)
    for cs in (False, True):
        p, st = bnb.quantize_nf4(W, compress_statistics=cs)
        out = torch.empty(4096, 4096, dtype=dt, device=dev)
        fn = lambda: bnb.dequantize_4bit(p, st, out=out)
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
        us = sorted(ev(gr.replay, 5) for _ in range(5))[2] / 20
        nbytes = 4096 * 4096 // 2 + 4096 * 64 * (1 if cs else 4) + 4096 * 4096 * W.element_size()
        print("%s double_quant=%s: %.2f us = %.0f GB/s" % (dt, cs, us, nbytes / us / 1e3), flush=True)
