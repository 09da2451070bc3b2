"""What the side work of k_gemm_dense's k-step costs: builds of tools/exp/dense_exp.hip with GD_ABL = 0 / 1 (no LDS-DMA pieces) / 2 (no fragment reads) /
3 / 4 (no barriers) / 7 (MFMAs alone) and GD_STAMPS = 1: cycles of the k-loop of workgroup 17's four waves, 4096^3 bf16 (results are wrong by design)."""
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
names = {0: "full k-step", 1: "no LDS-DMA pieces", 2: "no fragment reads", 3: "no pieces, no reads", 4: "no barriers", 7: "MFMAs alone"}
for v in (0, 1, 2, 3, 4, 7, 0):
    lib = ctypes.CDLL(os.path.join(here, f"libdense_abl{v}.so"))
    lib.exp_dense.restype = ctypes.c_int; lib.exp_dense.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
    lib.exp_dense_stamps.restype = ctypes.c_int; lib.exp_dense_stamps.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        assert lib.exp_dense(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, sp) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        lib.exp_dense(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, sp)
    e1.record(); e1.synchronize()
    host = (ctypes.c_ulonglong * 16)()
    assert lib.exp_dense_stamps(host) == 0
    tots = [host[4 * wv + 2] for wv in range(4)]
    print(f"{names[v]:22s}: k-loop {min(tots)}-{max(tots)} cycles = {sum(tots) / 4 / 64:7.1f} per k-step of 128 MFMAs; launch {e0.elapsed_time(e1) * 10:.2f} us", flush=True)
