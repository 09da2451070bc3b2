"""Linear8bit (W8A16) forward throughput at the 4096^3 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native
dev = "cuda"
lin = torch.nn.Linear(4096, 4096, bias=False).to(torch.bfloat16).to(dev)
l8 = bnb.Linear8bit.from_linear(lin)
x = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(10): l8(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): l8(x)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"Linear8bit 4096x4096 M=4096 bf16: {us:.1f} us, {2*4096**3/us/1e6:.0f} TFLOP/s ({_native.last_kernel()})")
