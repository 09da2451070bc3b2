"""Timing breakdown of the decode-beside path (tools/exp/libbeside_exp.so, built with GB_STAMPS): legs interleaved in one process +
the timeline of one call from the kernels' own s_memrealtime stamps.
ABL bits of the gated GEMM: 1 no wait for slab 0, 2 no spin at the in-loop gates, 4 no in-loop polls."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, functional as F
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbeside_exp.so"))
lib.exp_beside.restype = ctypes.c_int
lib.exp_beside.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(1)
W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16)
x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W, blocksize=64)
Wd = bnb.dequantize_4bit(packed, st).contiguous()
sync = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
nlib = _native.lib()
F.DECODE_BESIDE = False
ref = bnb.matmul_4bit(x, packed, st)


def run(mode, abl):
    rc = lib.exp_beside(mode, abl, x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), Wd.data_ptr(), sync.data_ptr(), out.data_ptr(), M, N, K, sp)
    assert rc == 0, rc


def dense():
    rc = nlib.mbnb_gemm_dense(x.data_ptr(), Wd.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, None, 0, 1 | (2 << 8), sp)
    assert rc == 0, rc


def zero():
    sync[:16384].zero_()


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


run(0, 0); torch.cuda.synchronize()
print("decoder + gated GEMM == two-launch:", torch.equal(out, ref), "sync words set:", int((sync[:16384] != 0).sum()), flush=True)
legs = {
    "dense alone": dense,
    "two launches (product)": lambda: bnb.matmul_4bit(x, packed, st),
    "decoder + gated, any-order": lambda: run(0, 0),
    "decoder, then gated (ordered)": lambda: run(3, 0),
    "two streams, decoder first": lambda: run(4, 0),
    "two streams, GEMM first": lambda: run(5, 0),
    "two streams, GEMM first, UNgated": lambda: (run(5, 7), zero()),
    "WaitValue": lambda: run(6, 0),
    "gated alone, polls, no waits": lambda: run(1, 3),
    "gated alone, no polls": lambda: run(1, 7),
    "decoder + UNgated any-order": lambda: (run(0, 7), zero()),
    "decoder alone (+ zeroing)": lambda: (run(2, 0), zero()),
    "zeroing alone": zero,
}
for f in legs.values():
    for _ in range(20):
        f()
torch.cuda.synchronize()
ev(dense, 3000)
res = {k: [] for k in legs}
for rep in range(7):
    for k, f in legs.items():
        res[k].append(ev(f, 100))
for k, v in res.items():
    v = sorted(v)
    print(f"{k:32s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}", flush=True)

def graph_time(fn, n=20):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        global sp
        sp_old = sp
        sp = side.cuda_stream
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        sp = sp_old
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[3]


zero(); torch.cuda.synchronize()
for name, fn in (("dense alone", dense), ("decoder, then gated (ordered)", lambda: run(3, 0)), ("two streams, decoder first", lambda: run(4, 0)),
                 ("two streams, GEMM first", lambda: run(5, 0))):
    try:
        print(f"HIP graph of 20 calls: {name:32s} {graph_time(fn):7.2f} us per call", flush=True)
    except Exception as e:
        print(f"HIP graph of 20 calls: {name:32s} failed: {e}", flush=True)
zero(); torch.cuda.synchronize()

def timeline(mode, label):
    zero(); torch.cuda.synchronize()
    for _ in range(6):
        run(mode, 0)
    torch.cuda.synchronize()
    s64 = sync.view(torch.int64)[2048:].cpu()
    dec = s64[: 12 * 256].view(256, 12).double() / 100.0          # us (100 MHz)
    gem = s64[4096: 4096 + 4 * 256].view(256, 4).double() / 100.0
    t0 = min(dec[:, 0].min().item(), gem[:, 0].min().item())
    print(f"---- {label}: timeline of the last of six back-to-back calls (us)")
    print(f"decoder: first start {dec[:,0].min().item()-t0:6.2f}  last start {dec[:,0].max().item()-t0:6.2f}  end (last) {dec[:,9].max().item()-t0:6.2f}")
    for u in range(8):
        print(f"  slab {u}: flag posted  first {dec[:,1+u].min().item()-t0:6.2f}  median {dec[:,1+u].median().item()-t0:6.2f}  last {dec[:,1+u].max().item()-t0:6.2f}")
    for name, c in (("start", 0), ("slab 0 seen", 1), ("loop end", 2), ("end", 3)):
        print(f"GEMM {name:12s}: first {gem[:,c].min().item()-t0:6.2f}  median {gem[:,c].median().item()-t0:6.2f}  last {gem[:,c].max().item()-t0:6.2f}")


timeline(0, "one stream, any-order")
timeline(4, "two streams, decoder first")
timeline(5, "two streams, GEMM first")
timeline(6, "WaitValue")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    run(6, 0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host time per WaitValue call (3 API calls): {(t1 - t0) / 200 * 1e6:.1f} us enqueue, {(t2 - t0) / 200 * 1e6:.1f} us incl. drain")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    run(3, 0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host time per ordered call (2 launches): {(t1 - t0) / 200 * 1e6:.1f} us enqueue, {(t2 - t0) / 200 * 1e6:.1f} us incl. drain")
