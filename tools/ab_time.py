"""A/B kernel time of the headline matmul_4bit (4096^3 NF4 bf16) under env switches, one subprocess per variant
(the switches are read once per process; MBNB_AB_SHAPE=M,N,K changes the shape).  usage: python tools/ab_time.py "" MBNB_VALUDEC=1 MBNB_NO_AM4=1 ..."""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
from mps_bitsandbytes_amd import synthetic
dev = "cuda"
M_, N_, K_ = (int(v) for v in os.environ.get("MBNB_AB_SHAPE", "4096,4096,4096").split(","))
W = synthetic.normal((N_, K_), torch.bfloat16, 1, 0.02).to(dev)
X = synthetic.normal((M_, K_), torch.bfloat16, 2, 1.0).to(dev)
packed, st = bnb.quantize_nf4(W)
ref = bnb.matmul_4bit(X[:256], packed, st).float()
for _ in range(100): out = bnb.matmul_4bit(X, packed, st)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): out = bnb.matmul_4bit(X, packed, st)
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
err = (out[:256].float() - ref).abs().max().item()
print(f"{min(ts):.2f} us  {_native.last_kernel()}  selfdiff={err}", flush=True)
'''
for v in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in v.split(","):
        if kv:
            k, x = kv.split("=")
            env[k] = x
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"[{v or 'default'}] {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
