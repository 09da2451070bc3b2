import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
N = K = M = 4096
W = torch.randn(N, K, device=dev).to(torch.bfloat16)
packed, st = bnb.quantize_nf4(W)
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
for _ in range(5):
    y = bnb.matmul_4bit(X, packed, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2048)()
lib = _native.lib()
rc = lib.mbnb_debug_read_stamps(buf)
import numpy as np
a = np.array(buf[:], dtype=np.uint64).reshape(2, 1024).astype(np.int64)
for s in range(2):
    t = a[s]
    d = np.diff(t)
    # events alternate: work-end, slot-start
    print("set", s, "first stamps deltas (cycles):")
    print(" ".join(str(int(v)) for v in d[40:104]))
