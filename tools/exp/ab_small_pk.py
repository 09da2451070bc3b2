"""A/B of GS_PK_MUL (gemm_small.h): the decode's two products per byte as ONE v_pk_mul_f32 against two v_mul_f32 -- libm0_pk{0,1}.so, k_gemm_small<bf16,
plain, MF, NF, 16> with one K slice: bit equality of the two builds (and against the library), then interleaved timing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
ml = [ctypes.CDLL(os.path.join(here, f"libm0_pk{v}.so")) for v in (0, 1)]
I64, P = ctypes.c_int64, ctypes.c_void_p
for l in ml:
    l.exp_small_v.restype = ctypes.c_int; l.exp_small_v.argtypes = [P] * 4 + [I64] * 3 + [P, ctypes.c_int]
sp = torch.cuda.current_stream().cuda_stream


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev); g.manual_seed(5)
VAR = {0: (8, 1), 1: (4, 1), 2: (2, 1), 3: (8, 2)}
for (M, N, K, var) in [(512, 4096, 4096, 0), (256, 4096, 4096, 0), (256, 4096, 4096, 1), (512, 4096, 4096, 3), (512, 8192, 2048, 3), (100, 4096, 1024, 1)]:
    W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    run = [lambda v=v: ml[v].exp_small_v(x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), outs[v].data_ptr(), M, N, K, sp, var) for v in range(2)]
    for r in run:
        assert r() == 0
    torch.cuda.synchronize()
    ref = bnb.matmul_4bit(x, packed, st)
    print(f"k_gemm_small<MF={VAR[var][0]}, NF={VAR[var][1]}> {M} x {N} x {K}: equal {torch.equal(outs[0], outs[1])}, max |diff| to the library {float((outs[1].float() - ref.float()).abs().max()):.3g}", flush=True)
    for r in run:
        for _ in range(30):
            r()
    ev(run[0], 1000)
    res = [[], []]
    for rep in range(9):
        for v in range(2):
            res[v].append(ev(run[v], 200))
    for v in range(2):
        r = sorted(res[v])
        print(f"  {'two v_mul_f32   ' if v == 0 else 'one v_pk_mul_f32'}: median {r[4]:7.2f} us  min {r[0]:7.2f}  max {r[-1]:7.2f}", flush=True)
