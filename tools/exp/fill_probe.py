import torch, statistics
dev=torch.device("cuda:0")
def graph_us(fn, n=20, reps=7):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        ts=[]
        for _ in range(reps):
            g.replay()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1)/n*1e3)
    return statistics.median(ts)
for mb in (8, 33.5, 134):
    n=int(mb*1e6)//2
    a=torch.empty(n,dtype=torch.bfloat16,device=dev); b=torch.empty(n,dtype=torch.bfloat16,device=dev)
    tz=graph_us(lambda: a.zero_()); tc=graph_us(lambda: b.copy_(a))
    print(f"{mb:6.1f} MB: zero_ {tz:7.2f} us = {mb/tz*1e-0:6.2f} TB/s written   copy_ {tc:7.2f} us = {2*mb/tc:6.2f} TB/s read+write", flush=True)
import sys, os
sys.path.insert(0, "/root/repo")
import mps_bitsandbytes_amd as bnb
W=torch.randn(4096,4096,device=dev).to(torch.bfloat16)
p,st=bnb.quantize_nf4(W)
out=torch.empty(4096,4096,dtype=torch.bfloat16,device=dev)
t=graph_us(lambda: bnb.dequantize_4bit(p,st,out=out) if False else bnb.dequantize_4bit(p,st))
print(f"dequantize_4bit 4096^2 (graph, incl. its torch.empty): {t:7.2f} us")
