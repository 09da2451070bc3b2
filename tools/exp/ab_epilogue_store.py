"""A/B of GD_EPI_STORE (common.h) in k_gemm_dense128 and k_gemm_i8_inplace: nontemporal (0) against write-through "sc1" (1) epilogue stores --
libm0_es{0,1}.so: bit equality, the kernel alone, and k_gemm_dense128 behind the (cached-store) dequantise pass as in a 1024-row matmul_4bit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
ml = [ctypes.CDLL(os.path.join(here, f"libm0_es{v}.so")) for v in (0, 1)]
dq = ctypes.CDLL(os.path.join(here, "libdq4_exp.so"))
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in ml:
    l.exp_d128.restype = ctypes.c_int; l.exp_d128.argtypes = [P] * 3 + [I64] * 3 + [P]
    l.exp_i8.restype = ctypes.c_int; l.exp_i8.argtypes = [P] * 5 + [I64] * 3 + [P]
dq.exp_dq4.restype = ctypes.c_int; dq.exp_dq4.argtypes = [ctypes.c_int] + [P] * 3 + [I64] * 2 + [P]
sp = torch.cuda.current_stream().cuda_stream
LAB = ["nontemporal", "sc1        "]


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, run):
    for v in range(2):
        for _ in range(30):
            run[v]()
    ev(run[0], 1500)
    res = [[], []]
    for rep in range(9):
        for v in range(2):
            res[v].append(ev(run[v], 200))
    print(name, flush=True)
    for v in range(2):
        r = sorted(res[v])
        print(f"  {LAB[v]}: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)


g = torch.Generator(device=dev); g.manual_seed(3)
for (M, N, K) in [(1024, 4096, 4096), (512, 4096, 4096), (1536, 4096, 4096)]:
    W = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    wd = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    assert dq.exp_dq4(1, packed.data_ptr(), st.absmax.data_ptr(), wd.data_ptr(), N, K, sp) == 0
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    run = [lambda v=v: ml[v].exp_d128(x.data_ptr(), wd.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)]
    for r in run:
        assert r() == 0
    torch.cuda.synchronize()
    print(f"k_gemm_dense128 {M} x {N} x {K} equal:", torch.equal(outs[0], outs[1]), flush=True)
    ab("  alone", run)

    def step(v):
        def f():
            dq.exp_dq4(1, packed.data_ptr(), st.absmax.data_ptr(), wd.data_ptr(), N, K, sp)
            ml[v].exp_d128(x.data_ptr(), wd.data_ptr(), outs[v].data_ptr(), M, N, K, sp)
        return f
    ab("  step: dequantise pass + k_gemm_dense128", [step(0), step(1)])
M = N = K = 4096
A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8); B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
sA = torch.rand(M, generator=g, device=dev) + 0.5; sB = torch.rand(N, generator=g, device=dev) + 0.5
outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
run = [lambda v=v: ml[v].exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)]
for r in run:
    assert r() == 0
torch.cuda.synchronize()
print("k_gemm_i8_inplace 4096^3 equal:", torch.equal(outs[0], outs[1]), flush=True)
ab("  alone", run)
