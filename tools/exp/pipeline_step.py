"""Dequantise-ahead: the weight of step i+1 decoded on a side stream (second scratch buffer) while step i's k_gemm_dense runs.
Every step still does both launches; only their order across steps changes (what a multi-layer model does with layer i+1's
weight).  Reports us per step against the in-order step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
M = N = K = 4096
dt = torch.bfloat16
W = torch.randn(N, K, generator=g, device=dev).to(dt)
p, st = bnb.quantize_nf4(W, blocksize=64)
X = torch.randn(M, K, generator=g, device=dev).to(dt)
lib, code = _native.lib(), _native.DTYPE_CODE[dt]
Wd = [torch.empty(N, K, dtype=dt, device=dev) for _ in range(2)]
Y = torch.empty(M, N, dtype=dt, device=dev)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def gemm(buf, stream):
    rc = lib.mbnb_gemm_dense(X.data_ptr(), buf.data_ptr(), code, None, code, Y.data_ptr(), M, N, K, K, None, 0, 1, stream.cuda_stream)
    assert rc == 0


def inorder(n):
    for i in range(n):
        bnb.dequantize_4bit(p, st, out=Wd[0])
        gemm(Wd[0], main)


def pipelined(n):
    # prologue: weight of step 0
    bnb.dequantize_4bit(p, st, out=Wd[0])
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    free = [torch.cuda.Event(), torch.cuda.Event()]
    ready[0].record(main)
    for i in range(n):
        cur, nxt = i & 1, (i + 1) & 1
        with torch.cuda.stream(side):
            if i >= 1:
                side.wait_event(free[nxt])          # the GEMM that last read Wd[nxt] (step i-1) has finished
            bnb.dequantize_4bit(p, st, out=Wd[nxt])   # weight of step i+1
            ready[nxt].record(side)
        main.wait_event(ready[cur])
        gemm(Wd[cur], main)
        free[cur].record(main)
    main.wait_stream(side)


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(main)
    fn(n)
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(3):
    inorder(500)
ref = bnb.matmul_4bit(X, p, st)
pipelined(4)
torch.cuda.synchronize()
assert torch.equal(Y, ref)
a = sorted(ev(inorder, 200) for _ in range(5))[2]
b = sorted(ev(pipelined, 200) for _ in range(5))[2]
print("in-order %.2f us/step (%.0f TFLOP/s), dequantise-ahead %.2f us/step (%.0f TFLOP/s)" % (a, 2.0 * M * N * K / a / 1e6, b, 2.0 * M * N * K / b / 1e6))
