"""A/B of GD_M0_GROUP (common.h): ONE M0 write per four LDS-DMA pieces (the instruction offset carries the piece inside the group) against one
per piece, kernel by kernel (libdense_exp{0,1}.so, libm0_exp{0,1}.so): bit equality of the two builds, then interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
dl = [ctypes.CDLL(os.path.join(here, f"libdense_exp{v}.so")) for v in (0, 1)]
ml = [ctypes.CDLL(os.path.join(here, f"libm0_exp{v}.so")) for v in (0, 1)]
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in dl:
    l.exp_dense.restype = ctypes.c_int; l.exp_dense.argtypes = [P] * 3 + [I64] * 3 + [P]
for l in ml:
    l.exp_d128.restype = ctypes.c_int; l.exp_d128.argtypes = [P] * 3 + [I64] * 3 + [P]
    l.exp_i8.restype = ctypes.c_int; l.exp_i8.argtypes = [P] * 5 + [I64] * 3 + [P]
    l.exp_small.restype = ctypes.c_int; l.exp_small.argtypes = [P] * 4 + [I64] * 3 + [P]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, run, outs):
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
        print(f"  {'one M0 write per piece   ' if v == 0 else 'one M0 write per 4 pieces'}: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)


g = torch.Generator(device=dev); g.manual_seed(3)
for (M, N, K) in [(4096, 4096, 4096), (4096, 11008, 4096)]:
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ab(f"k_gemm_dense {M} x {N} x {K}", [lambda v=v: dl[v].exp_dense(x.data_ptr(), w.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs)
for (M, N, K) in [(1024, 4096, 4096), (512, 4096, 4096)]:
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ab(f"k_gemm_dense128 {M} x {N} x {K}", [lambda v=v: ml[v].exp_d128(x.data_ptr(), w.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs)
M = N = K = 4096
A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8); B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
sA = torch.rand(M, generator=g, device=dev) + 0.5; sB = torch.rand(N, generator=g, device=dev) + 0.5
outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
ab("k_gemm_i8_inplace 4096^3", [lambda v=v: ml[v].exp_i8(A.data_ptr(), B.data_ptr(), sA.data_ptr(), sB.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs)
for (M, N, K) in [(512, 4096, 4096), (300, 4096, 2048)]:
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    ab(f"k_gemm_small (16 steps) {M} x {N} x {K}", [lambda v=v: ml[v].exp_small(x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), outs[v].data_ptr(), M, N, K, sp) for v in range(2)], outs)
    print("   vs library:", torch.equal(outs[1], bnb.matmul_4bit(x, packed, st)))
