"""Attainable HBM read bandwidth on this box (torch reductions / copies over 1 GiB), for calibrating the GEMV roofline."""
import torch, sys
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (9, 64, 1024):
    t = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    ts = [torch.empty_like(t).copy_(t) for _ in range(max(1, 1024 // mb))] if mb < 1024 else [t]
    i = [0]
    def rd():
        i[0] = (i[0] + 1) % len(ts); return ts[i[0]].sum()
    s = timeit(rd)
    print(f"{mb} MiB sum (rotating {len(ts)} buffers): {t.numel()*4/s/1e9:.0f} GB/s read", flush=True)
    o = torch.empty_like(t)
    s = timeit(lambda: o.copy_(t))
    print(f"{mb} MiB copy: {2*t.numel()*4/s/1e9:.0f} GB/s read+write", flush=True)
