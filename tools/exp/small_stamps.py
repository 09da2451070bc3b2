"""Cycle stamps of one wave of k_gemm_small (16-step form) at 512 x 4096 x 4096 and 128 x 4096 x 4096: per step, the wait for the step's LDS-DMA
(vmcnt(0)), the barrier, and the 64 MFMAs + decode + next step's piece issue."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mps_bitsandbytes_amd as bnb
dev = torch.device("cuda:0")
abl = sys.argv[1] if len(sys.argv) > 1 else ""
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libsmall_stamps{abl}.so"))
lib.exp_small_stamps.restype = ctypes.c_int
lib.exp_small_stamps.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
for (M, N, K) in [(512, 4096, 4096), (128, 4096, 4096), (128, 4096, 2048)]:
    W = torch.randn(N, K, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    packed, st = bnb.quantize_nf4(W, blocksize=64)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    host = (ctypes.c_ulonglong * 256)()
    for rep in range(3):
        rc = lib.exp_small_stamps(x.data_ptr(), packed.data_ptr(), st.absmax.data_ptr(), out.data_ptr(), M, N, K, host, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    steps = K // 256
    print(f"{M} x {N} x {K}: {steps} steps; equal to the library: {torch.equal(out, bnb.matmul_4bit(x, packed, st)) if M > 256 else 'n/a (split in the library)'}")
    tot = {"wait": 0, "barrier": 0, "compute": 0, "between": 0}
    for t in range(steps):
        ta, tw, tb, tc = host[4 * t], host[4 * t + 1], host[4 * t + 2], host[4 * t + 3]
        nxt = host[4 * (t + 1)] if t + 1 < steps else tc
        if not abl: print(f"  step {t:2d}: vmcnt wait {tw - ta:6d}  barrier {tb - tw:6d}  compute {tc - tb:6d} cycles   (to next step {nxt - tc})")
        tot["wait"] += tw - ta; tot["barrier"] += tb - tw; tot["compute"] += tc - tb; tot["between"] += nxt - tc
    print(f"  abl={abl or 0} totals:", tot, "   whole loop:", host[4 * (steps - 1) + 3] - host[0], "cycles")
