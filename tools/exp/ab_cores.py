"""Probe (round 3): does a small helper kernel (<= 32 VGPRs, 4 waves per workgroup, one workgroup per CU) run BESIDE k_gemm_dense
(476 of a SIMD's 512 registers) when the two are launched on two streams with a fork / join of events, and what does the
fork / join cost?  Prints the time per iteration of: the GEMM alone; the helper alone; fork -> (helper | GEMM) -> join in both
launch orders; the same with a 64-register helper (must not fit beside the GEMM); with a helper that streams 8 MB -> 32 MB like the
dequantise pass; with an empty helper (the price of the fork / join itself)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mps_bitsandbytes_amd import _native
dev = torch.device("cuda:0")
lib = _native.lib()
hl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcores_exp.so"))
hl.cores_helper.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
dt = torch.bfloat16
M = N = K = 4096
gen = torch.Generator(device=dev); gen.manual_seed(5)
x = torch.randn(M, K, generator=gen, device=dev).to(dt)
w = (torch.randn(N, K, generator=gen, device=dev) * 0.05).to(dt)
out = torch.empty(M, N, dtype=dt, device=dev)
src = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
dst = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
stamps = torch.zeros(1024 * 6, dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm(st):
    rc = lib.mbnb_gemm_dense(x.data_ptr(), w.data_ptr(), 1, None, 1, out.data_ptr(), M, N, K, K, None, 0, 1 | (2 << 8), st.cuda_stream)
    assert rc == 0, (rc, lib.mbnb_last_error())


def helper(st, regs, grid, iters, work):
    rc = hl.cores_helper(regs, grid, iters, work, src.data_ptr(), dst.data_ptr(), (32 << 20) // 16 // grid, stamps.data_ptr(), st.cuda_stream)
    assert rc == 0, rc


def timed(body, n=20, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(n):
            body()
        e1.record(s1)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


def forkjoin(first, regs, grid, iters, work):
    def body():
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)
        if first == "helper":
            helper(s2, regs, grid, iters, work); gemm(s1)
        else:
            gemm(s1); helper(s2, regs, grid, iters, work)
        j = torch.cuda.Event(); j.record(s2); s1.wait_event(j)
    return body


def placement(grid):
    torch.cuda.synchronize()
    s = stamps[: 6 * grid].view(grid, 6).cpu()
    span = (s[:, 1].max() - s[:, 0].min()).item() / 100.0
    each = ((s[:, 1] - s[:, 0]).double().mean().item()) / 100.0
    key = (s[:, 3] & 0xF) * 4096 + ((s[:, 2] >> 8) & 0x7FF)        # xcc, (se, sh, cu)
    cus, cnt = torch.unique(key, return_counts=True)
    return f"span {span:6.1f} us, mean per workgroup {each:6.1f} us, {len(cus)} distinct CUs, at most {cnt.max().item()} per CU"


for _ in range(50):
    gemm(s1)
torch.cuda.synchronize()
print(f"GEMM alone                       {timed(lambda: gemm(s1)):7.2f} us", flush=True)
for iters in (2000, 8000):
    print(f"helper alone (VALU x{iters}), 256 wg  {timed(lambda: helper(s1, 32, 256, iters, 0)):7.2f} us   {placement(256)}", flush=True)
print(f"helper alone (32 MB stream), 256 wg {timed(lambda: helper(s1, 32, 256, 0, 1)):7.2f} us   {placement(256)}", flush=True)
print(f"GEMM, then helper(8000) in ONE stream {timed(lambda: (gemm(s1), helper(s1, 32, 256, 8000, 0))):7.2f} us", flush=True)
for first in ("helper", "gemm"):
    for regs in (32, 64):
        for (iters, work, name) in ((0, 0, "empty"), (2000, 0, "VALU x2000"), (8000, 0, "VALU x8000"), (0, 1, "32 MB stream")):
            t = timed(forkjoin(first, regs, 256, iters, work))
            print(f"fork/join, {first:6s} first, helper {regs} regs, {name:12s}: {t:7.2f} us per iteration   helper: {placement(256)}", flush=True)
