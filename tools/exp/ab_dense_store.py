"""A/B of GD_EPI_STORE (common.h): cache policy of k_gemm_dense's 16-bit epilogue stores (libdense_es{0..4}.so), the kernel alone and inside the
step (write-through dequantise pass of tools/exp/libdq4_exp.so variant 22 + the GEMM): bit equality, then interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
V = (0, 1, 2, 3, 4)
NAMES = {0: "nt (product)", 1: "sc1", 2: "sc1 nt", 3: "sc0 sc1 nt", 4: "plain"}
dl = [ctypes.CDLL(os.path.join(here, f"libdense_es{v}.so")) for v in V]
dq = ctypes.CDLL(os.path.join(here, "libdq4_exp.so"))
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in dl:
    l.exp_dense.restype = ctypes.c_int; l.exp_dense.argtypes = [P] * 3 + [I64] * 3 + [P]
dq.exp_dq4.restype = ctypes.c_int; dq.exp_dq4.argtypes = [ctypes.c_int] + [P] * 3 + [I64] * 2 + [P]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(title, run):
    for r in run:
        for _ in range(30):
            r()
    ev(run[0], 1000)
    res = [[] for _ in V]
    for rep in range(9):
        for i in range(len(V)):
            res[i].append(ev(run[i], 200))
    print(title, flush=True)
    for i, v in enumerate(V):
        r = sorted(res[i])
        print(f"  {NAMES[v]:14s}: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)


g = torch.Generator(device=dev); g.manual_seed(3)
for (M, N, K) in [(4096, 4096, 4096), (4096, 11008, 4096)]:
    W = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    wd = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    assert dq.exp_dq4(22, packed.data_ptr(), st.absmax.data_ptr(), wd.data_ptr(), N, K, sp) == 0
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in V]
    run = [lambda i=i: dl[i].exp_dense(x.data_ptr(), wd.data_ptr(), outs[i].data_ptr(), M, N, K, sp) for i in range(len(V))]
    for r in run:
        assert r() == 0
    torch.cuda.synchronize()
    print(f"{M} x {N} x {K} equal:", [torch.equal(outs[0], o) for o in outs], "to the library:", torch.equal(outs[0], bnb.matmul_4bit(x, packed, st)), flush=True)
    ab("  k_gemm_dense alone", run)

    def step(i):
        def f():
            dq.exp_dq4(22, packed.data_ptr(), st.absmax.data_ptr(), wd.data_ptr(), N, K, sp)
            dl[i].exp_dense(x.data_ptr(), wd.data_ptr(), outs[i].data_ptr(), M, N, K, sp)
        return f
    ab("  step: write-through dequantise pass + k_gemm_dense", [step(i) for i in range(len(V))])
