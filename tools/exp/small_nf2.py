"""k_gemm_small with 128 weight rows per workgroup (NF = 2) against 64 (NF = 1) on the round-3 pipeline: cycles per step of one wave (stamps) and the call time, 512 x 4096 x 2048
(8 steps, one slice: NF = 1 -> 256 workgroups, NF = 2 -> 128)."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmall_stamps.so"))
for f in (lib.exp_small_stamps, lib.exp_small_stamps_nf2):
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
M, N, K = 512, 4096, 2048
W = torch.randn(N, K, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W, blocksize=64)
ref = bnb.matmul_4bit(x, packed, st)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
host = (ctypes.c_ulonglong * 256)()
for name, f in (("NF = 1 (64 weight rows per workgroup, 256 workgroups)", lib.exp_small_stamps), ("NF = 2 (128 rows, 128 workgroups)", lib.exp_small_stamps_nf2)):
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), out.data_ptr(), M, N, K, host, torch.cuda.current_stream().cuda_stream)
        e1.record(); e1.synchronize()
        assert rc == 0, rc
        ts.append(e0.elapsed_time(e1) * 1e3)
    steps = K // 256
    comp = [host[4 * t + 3] - host[4 * t + 2] for t in range(steps)]
    wait = [host[4 * t + 1] - host[4 * t] for t in range(steps)]
    print(f"{name}: equal to the library {torch.equal(out, ref)}; compute cycles per step {sum(comp) / steps:.0f} (wait {sum(wait) / steps:.0f}); loop {host[4 * (steps - 1) + 3] - host[0]} cycles")
