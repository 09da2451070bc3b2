"""Where do the waits of k_gemm_dense's k-step go?  Four builds of tools/exp/dense_exp.hip (GD_STAMPS = 1..4) each time ONE wait or barrier with a
pair of s_memtime stamps and sum it over the 64 k-steps of workgroup 17's four waves (4096^3 bf16)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
names = {1: "vmcnt(16) wait of barrier 2", 2: "barrier 2", 3: "lgkmcnt(0) wait of barrier 1", 4: "barrier 1"}
for v in (1, 2, 3, 4):
    lib = ctypes.CDLL(os.path.join(here, f"libdense_st{v}.so"))
    lib.exp_dense.restype = ctypes.c_int; lib.exp_dense.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
    lib.exp_dense_stamps.restype = ctypes.c_int; lib.exp_dense_stamps.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        assert lib.exp_dense(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, sp) == 0
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * 16)()
    assert lib.exp_dense_stamps(host) == 0
    for wv in range(4):
        s, c, tot = host[4 * wv], host[4 * wv + 1], host[4 * wv + 2]
        print(f"{names[v]:30s} wave {wv}: {s / max(c, 1):7.1f} cycles per k-step ({c} stamps), loop {tot} cycles = {tot / 64:.0f} per k-step", flush=True)
