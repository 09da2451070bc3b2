"""The waits of k_gemm_dense128's k-step (32 MFMAs per wave) timed in place: libd128_st1.so (vmcnt(8) wait) / libd128_st2.so (barrier), 1024 x 4096 x 4096 and
512 x 4096 x 4096 bf16, workgroup 17's four waves."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
sp = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(1024, 4096, 4096), (512, 4096, 4096)]:
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for v, name in ((1, "vmcnt(8) wait"), (2, "barrier"), ("abl1", "no pieces"), ("abl2", "no frag reads"), ("abl3", "neither")):
        lib = ctypes.CDLL(os.path.join(here, f"libd128_st{v}.so" if isinstance(v, int) else f"libd128_{v}.so"))
        lib.exp_d128.restype = ctypes.c_int; lib.exp_d128.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
        lib.exp_d128_stamps.restype = ctypes.c_int; lib.exp_d128_stamps.argtypes = [ctypes.c_void_p]
        for _ in range(20):
            assert lib.exp_d128(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, sp) == 0
        torch.cuda.synchronize()
        host = (ctypes.c_ulonglong * 16)()
        assert lib.exp_d128_stamps(host) == 0
        for wv in range(4):
            s, c, tot = host[4 * wv], host[4 * wv + 1], host[4 * wv + 2]
            print(f"{M} x {N} x {K}  {name:14s} wave {wv}: {s / max(c, 1):7.1f} cycles per k-step ({c} stamps), loop {tot} cycles = {tot / 64:.0f} per k-step", flush=True)
