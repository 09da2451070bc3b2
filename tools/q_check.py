"""Parity + time of the four-wave 256x256 kernel (MBNB_Q4W=1) against the production eight-wave kernel on the same inputs:
outputs must be bit-identical (same decoded B bits, same accumulation order per element)."""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic
dev = "cuda"
res = []
for (M, N, K, dt, cs, bias) in [(4096, 4096, 4096, torch.bfloat16, False, False), (2500, 2600, 512, torch.float16, True, True),
                                 (4096, 11008, 4096, torch.bfloat16, True, False), (2304, 3000, 256, torch.float16, False, True)]:
    W = synthetic.normal((N, K), dt, 1, 0.02).to(dev)
    X = synthetic.normal((M, K), dt, 2, 1.0).to(dev)
    b = synthetic.normal((N,), dt, 3, 1.0).to(dev) if bias else None
    packed, st = bnb.quantize_4bit(W, compress_statistics=cs, quant_type="nf4")
    for _ in range(20): out = bnb.matmul_4bit(X, packed, st, b)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): out = bnb.matmul_4bit(X, packed, st, b)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 100 * 1e3)
    res.append((M, N, K, min(ts), _native.last_kernel(), out.view(torch.int16).cpu()))
torch.save(res, sys.argv[1])
'''
outs = {}
for tag, env_add in (("p", {}), ("q", {"MBNB_Q4W": "1"})):
    env = dict(os.environ); env.update(env_add)
    path = f"/tmp/qcheck_{tag}.pt"
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True)
    if r.returncode:
        print(tag, "FAILED", r.stderr[-800:]); sys.exit(1)
    import torch
    outs[tag] = torch.load(path)
for a, b in zip(outs["p"], outs["q"]):
    same = bool((a[5] == b[5]).all())
    print(f"M={a[0]} N={a[1]} K={a[2]}: {a[4]} {a[3]:.1f} us | {b[4]} {b[3]:.1f} us | bit-identical={same}", flush=True)
