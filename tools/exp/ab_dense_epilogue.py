"""A/B of GD_EPI_GROUPS (common.h): k_gemm_dense's 16-bit epilogue staged through the wave's LDS in parts of 64 / 32 / 16 rows
(libdense_ep{4,2,1}.so): bit equality of the builds, then interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
V = (4, 2, 1)
dl = [ctypes.CDLL(os.path.join(here, f"libdense_ep{v}.so")) for v in V]
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in dl:
    l.exp_dense.restype = ctypes.c_int; l.exp_dense.argtypes = [P] * 3 + [I64] * 3 + [P]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev); g.manual_seed(3)
for (M, N, K) in [(4096, 4096, 4096), (4096, 11008, 4096), (4000, 4100 // 8 * 8, 1024)]:
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in V]
    run = [lambda i=i: dl[i].exp_dense(x.data_ptr(), w.data_ptr(), outs[i].data_ptr(), M, N, K, sp) for i in range(len(V))]
    for r in run:
        assert r() == 0
    torch.cuda.synchronize()
    print(f"k_gemm_dense {M} x {N} x {K} equal:", [torch.equal(outs[0], o) for o in outs], "finite:", bool(torch.isfinite(outs[0].float()).all()), flush=True)
    for r in run:
        for _ in range(30):
            r()
    ev(run[0], 1000)
    res = [[] for _ in V]
    for rep in range(9):
        for i in range(len(V)):
            res[i].append(ev(run[i], 200))
    for i, v in enumerate(V):
        r = sorted(res[i])
        print(f"  parts of {16 * v:2d} rows: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)
