"""A/B of the 16-bit epilogues' part sizes: k_gemm_dense128 one part of 64 rows against four of 16 (G128_EPI_ONE_PART), k_gemm_i8_inplace four
parts of 32 rows against eight of 16 (GI8_EPI_PARTS) -- libm0_ep{0,1}.so: bit equality of the two builds, then interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
ml = [ctypes.CDLL(os.path.join(here, f"libm0_ep{v}.so")) for v in (0, 1)]
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in ml:
    l.exp_d128.restype = ctypes.c_int; l.exp_d128.argtypes = [P] * 3 + [I64] * 3 + [P]
    l.exp_i8.restype = ctypes.c_int; l.exp_i8.argtypes = [P] * 5 + [I64] * 3 + [P]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, run, outs, labels):
    for v in range(2):
        assert run[v]() == 0
    torch.cuda.synchronize()
    print(name, "equal:", torch.equal(outs[0], outs[1]), "finite:", bool(torch.isfinite(outs[1].float()).all()), flush=True)
    for v in range(2):
        for _ in range(30):
            run[v]()
    ev(run[0], 1500)
    res = [[], []]
    for rep in range(9):
        for v in range(2):
            res[v].append(ev(run[v], 200))
    for v in range(2):
        r = sorted(res[v])
        print(f"  {labels[v]}: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)


g = torch.Generator(device=dev); g.manual_seed(3)
for (M, N, K) in [(1024, 4096, 4096), (512, 4096, 4096), (1000, 4104, 1024)]:
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ab(f"k_gemm_dense128 {M} x {N} x {K}", [lambda v=v: ml[v].exp_d128(x.data_ptr(), w.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs,
       ["one part of 64 rows ", "four parts of 16 rows"])
for (M, N, K) in [(4096, 4096, 4096), (4096, 4096, 1024)]:
    A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8); B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
    sA = torch.rand(M, generator=g, device=dev) + 0.5; sB = torch.rand(N, generator=g, device=dev) + 0.5
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ab(f"k_gemm_i8_inplace {M} x {N} x {K}", [lambda v=v: ml[v].exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs,
       ["four parts of 32 rows ", "eight parts of 16 rows"])
