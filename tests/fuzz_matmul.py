"""Randomised parity sweep of matmul_4bit against the CPU oracle (oracle/: checker only): `python tests/fuzz_matmul.py [cases] [seed]` (test infrastructure, not collected by pytest) draws shapes around
every dispatch edge of csrc/matmul4_kernels.hip (rows 1 .. 4300, ragged and aligned N / K, blocksizes 32-2048, both code tables, plain and nested absmax,
bias, f16 / bf16 / f32 outputs), checks the packed bytes bit for bit and the product within the suite's tolerance, and prints which kernel served each
case.  Exit status 1 on the first failure (the case is printed so that it can be added to tests/test_gpu_parity.py).  `--int8 [cases] [seed]`: the same for matmul_int8 and linear_int8 (W8A16).""" 
import os, sys, random, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic
import oracle
DEV = torch.device("cuda:0")
TOL = {torch.float16: 2e-3, torch.bfloat16: 1e-2, torch.float32: 2e-3}


def rel_fro(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def draw(rng):
    edge_m = [1, 2, 16, 17, 28, 29, 32, 33, 64, 65, 128, 129, 192, 256, 257, 320, 384, 385, 512, 513, 640, 1024, 1025, 2047, 2048, 2049]
    M = rng.choice(edge_m) if rng.random() < 0.5 else (rng.randint(1, 700) if rng.random() < 0.8 else rng.randint(700, 4300))
    kind = rng.random()
    if kind < 0.45:
        K = 256 * rng.randint(1, 24)
    elif kind < 0.8:
        K = 64 * rng.randint(1, 48)
    else:
        K = 8 * rng.randint(4, 200)
    N = rng.choice([64, 256, 512, 1000, 1024, 2048, 4096, 520, 777, 11008 // 4, 4104, 5120, 6144]) if rng.random() < 0.7 else rng.randint(16, 3000)   # 5120 / 6144 x >= 3841 rows: the column-balanced grids of round 4
    while M * N * K > 6e9:      # the oracle's share of the run
        M = max(1, M // 2)
    bs = rng.choice([64] * 5 + [32, 128, 256, 2048])
    if K % bs:
        bs = 64 if K % 64 == 0 else 32 if K % 32 == 0 else 8
    return dict(M=M, N=N, K=K, dt=rng.choice([torch.float16, torch.bfloat16]), qt=rng.choice(["nf4", "nf4", "fp4"]), bs=bs, cs=rng.random() < 0.3,
                bias=rng.random() < 0.5, cd=rng.choice([None, None, None, torch.float32, torch.float16, torch.bfloat16]))


def run(c, seed):
    M, N, K, dt = c["M"], c["N"], c["K"], c["dt"]
    W = synthetic.normal((N, K), dt, seed=seed)
    X = synthetic.normal((M, K), dt, seed=seed + 1)
    b = synthetic.normal((N,), dt, seed=seed + 2) if c["bias"] else None
    o_packed, o_absmax, o_st2 = oracle.quantize_4bit(W, c["bs"], c["qt"], c["cs"])
    packed, st = bnb.quantize_4bit(W.to(DEV), blocksize=c["bs"], quant_type=c["qt"], compress_statistics=c["cs"])
    assert torch.equal(packed.cpu(), o_packed), "packed bytes differ"
    y = bnb.matmul_4bit(X.to(DEV), packed, st, None if b is None else b.to(DEV), c["cd"])
    kern = _native.last_kernel()
    y_ref = oracle.matmul_4bit(X, o_packed, o_absmax, (N, K), c["bs"], c["qt"], dt, b, c["cd"], o_st2)
    assert y.dtype == y_ref.dtype and tuple(y.shape) == tuple(y_ref.shape), "dtype / shape"
    assert bool(torch.isfinite(y).all()), "non-finite output"
    err = rel_fro(y, y_ref)
    tol = max(TOL[dt], TOL[y.dtype])
    assert err <= tol, f"rel-err {err:.3e} > {tol}"
    y2 = bnb.matmul_4bit(X.to(DEV), packed, st, None if b is None else b.to(DEV), c["cd"])
    assert torch.equal(y, y2), "not run-to-run deterministic"
    return kern, err


def draw8(rng):
    M = rng.choice([1, 8, 16, 17, 32, 33, 64, 128, 129, 256, 257, 384, 385, 512, 1024, 2048, 2500, 4096]) if rng.random() < 0.5 else rng.randint(1, 3000)
    K = rng.choice([128, 256, 384, 512, 1024, 2048, 4096]) if rng.random() < 0.7 else 8 * rng.randint(2, 300)
    N = rng.choice([16, 64, 256, 512, 1000, 1024, 2048, 2600, 4096]) if rng.random() < 0.7 else rng.randint(8, 3000)
    while M * N * K > 6e9:
        M = max(1, M // 2)
    return dict(op=rng.choice(["matmul_int8", "linear_int8", "linear_int8"]), M=M, N=N, K=K, dt=rng.choice([torch.float16, torch.bfloat16]), bias=rng.random() < 0.5)


def run8(c, seed):
    M, N, K, dt = c["M"], c["N"], c["K"], c["dt"]
    if c["op"] == "matmul_int8":
        A = synthetic.int8_tensor((M, K), seed=seed); B = synthetic.int8_tensor((K, N), seed=seed + 1)
        sa = synthetic.normal((M,), torch.float32, seed=seed + 2).abs() + 0.5; sb = synthetic.normal((N,), torch.float32, seed=seed + 3).abs() + 0.5
        y = bnb.matmul_int8(A.to(DEV), B.to(DEV), sa.to(DEV), sb.to(DEV), dt)
        kern = _native.last_kernel()
        y_ref = oracle.matmul_int8(A, B, sa, sb, dt)
        y2 = bnb.matmul_int8(A.to(DEV), B.to(DEV), sa.to(DEV), sb.to(DEV), dt)
    else:
        W = synthetic.normal((N, K), dt, seed=seed); X = synthetic.normal((M, K), dt, seed=seed + 1)
        b = synthetic.normal((N,), dt, seed=seed + 2) if c["bias"] else None
        oq, osc = oracle.quantize_rowwise(W)
        q, sc = bnb.quantize_rowwise(W.to(DEV))
        assert torch.equal(q.cpu(), oq) and torch.equal(sc.cpu().float(), osc.float()), "quantize_rowwise differs"
        y = bnb.linear_int8(X.to(DEV), q, sc, None if b is None else b.to(DEV))
        kern = _native.last_kernel()
        y_ref = oracle.linear_int8(X, oq, osc, b)
        y2 = bnb.linear_int8(X.to(DEV), q, sc, None if b is None else b.to(DEV))
    assert y.dtype == y_ref.dtype and tuple(y.shape) == tuple(y_ref.shape), "dtype / shape"
    finite = torch.isfinite(y_ref)
    assert torch.equal(torch.isfinite(y.cpu()), finite), "finiteness differs"
    err = rel_fro(torch.where(finite, y.cpu(), torch.zeros_like(y_ref)), torch.where(finite, y_ref, torch.zeros_like(y_ref)))
    assert err <= TOL[dt], f"rel-err {err:.3e} > {TOL[dt]}"
    assert torch.equal(y, y2), "not run-to-run deterministic"
    return kern, err


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--int8":
        cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
        seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
        rng = random.Random(seed0)
        served = collections.Counter()
        t0 = time.time()
        for i in range(cases):
            c = draw8(rng)
            try:
                kern, err = run8(c, 1000 * seed0 + 7 * i)
            except Exception as e:      # noqa: BLE001 -- report the case, then fail
                print(f"FAILED case {i}: {c}: {type(e).__name__}: {e} (kernel {_native.last_kernel()})", flush=True)
                sys.exit(1)
            served[kern] += 1
            print(f"{i:4d} {c['op']:12s} M={c['M']:5d} N={c['N']:5d} K={c['K']:5d} {str(c['dt'])[6:]:8s} bias={int(c['bias'])} {kern:22s} rel-err {err:.2e}  [{time.time() - t0:5.0f} s]", flush=True)
        print("all", cases, "8-bit cases within tolerance; served by:", dict(served), flush=True)
        sys.exit(0)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed0)
    served = collections.Counter()
    t0 = time.time()
    for i in range(cases):
        c = draw(rng)
        try:
            kern, err = run(c, 1000 * seed0 + 7 * i)
        except Exception as e:      # noqa: BLE001 -- report the case, then fail
            print(f"FAILED case {i}: {c}: {type(e).__name__}: {e} (kernel {_native.last_kernel()})", flush=True)
            sys.exit(1)
        served[kern] += 1
        print(f"{i:4d} M={c['M']:5d} N={c['N']:5d} K={c['K']:5d} {str(c['dt'])[6:]:8s} {c['qt']} bs={c['bs']:4d} cs={int(c['cs'])} bias={int(c['bias'])} cd={str(c['cd'])[6:] if c['cd'] else '-':8s} "
              f"{kern:22s} rel-err {err:.2e}  [{time.time() - t0:5.0f} s]", flush=True)
    print("all", cases, "cases within tolerance; served by:", dict(served), flush=True)
