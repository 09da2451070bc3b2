#!/usr/bin/env python3
"""
Golden vectors for FP8 E4M3: quantize_fp8_e4m3 / dequantize_fp8_e4m3 / matmul_fp8_e4m3 / LinearFP8,
captured by RUNNING THE REFERENCE's Python CPU path here (data only).  Includes the adversarial rows that pin the
reference's exponent rule (floor(torch.log2(v)) flips to k a few f32 ulps below 2^k) and its clamps.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fp8.py     -> g7_fp8.npz, manifest_fp8.json
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

import mps_bitsandbytes as ref  # noqa: E402
from mps_bitsandbytes.nn import LinearFP8  # noqa: E402
from mps_bitsandbytes_amd import synthetic  # noqa: E402
from make_golden import bits, DT  # noqa: E402


def adversarial_rows():
    """Rows whose absmax is exactly 448 (scale 1.0) so the normalized values ARE the listed values: every f32 within
    8 ulps of 2^k for k in [-9, 9], mantissa rounding midpoints, the clamp edges, zeros of both signs, tiny values."""
    rows = []
    for k in range(-9, 10):
        base = np.float32(2.0 ** k).view(np.int32)
        vals = (np.arange(base - 8, base + 9, dtype=np.int32)).view(np.float32)
        rows.append(vals)
        rows.append(-vals)
    mids = []
    for k in range(-6, 9):
        for m in range(8):
            v = np.float32((1 + (m + 0.5) / 8) * 2.0 ** k)          # halfway between two FP8 mantissas
            b = v.view(np.int32)
            mids.append((np.arange(b - 2, b + 3, dtype=np.int32)).view(np.float32))
    rows.append(np.concatenate(mids)[:17 * 8].reshape(8, 17).reshape(-1)[:17])
    out = []
    for r in rows:
        r = np.asarray(r, dtype=np.float32)[:17]
        row = np.zeros(24, dtype=np.float32)
        row[:len(r)] = r
        row[-1] = 448.0                                               # pins scale = 1.0
        row[-2], row[-3], row[-4] = -0.0, 1e-30, 300.0
        out.append(row)
    allm = np.concatenate(mids)
    for i in range(0, len(allm) - 23, 23):
        row = np.zeros(24, dtype=np.float32)
        row[:23] = allm[i:i + 23]
        row[-1] = 448.0
        out.append(row)
    return torch.from_numpy(np.stack(out))


def main():
    arrays, cases = {}, []
    # quantize / dequantize
    inputs = [("adv", adversarial_rows()), ("f16", synthetic.normal((64, 128), torch.float16, seed=900)),
              ("bf16", synthetic.normal((33, 70), torch.bfloat16, seed=901, std=3.0)),
              ("f32", synthetic.normal((16, 256), torch.float32, seed=902, std=0.02))]
    z = torch.zeros(4, 16); z[1, 3] = 5.0; z[2, :] = -1e-20
    inputs.append(("zeros", z))
    for name, x in inputs:
        q, s = ref.quantize_fp8_e4m3(x)
        arrays[f"q_{name}_x"], arrays[f"q_{name}_q"], arrays[f"q_{name}_s"] = bits(x), bits(q), bits(s)
        for dt in ("f16", "bf16", "f32"):
            arrays[f"q_{name}_deq_{dt}"] = bits(ref.dequantize_fp8_e4m3(q, s, DT[dt]))
        cases.append(dict(kind="quant", name=name, shape=list(x.shape), dtype={torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[x.dtype]))
    # every byte through the decoder
    allb = torch.arange(256, dtype=torch.uint8).reshape(2, 128)
    arrays["dec_all"] = bits(ref.dequantize_fp8_e4m3(allb, torch.tensor([1.0, 0.37]), torch.float32))
    # matmul_fp8_e4m3 / LinearFP8
    for ci, (M, K, N, dt, has_bias) in enumerate([(4, 64, 32, "f16", True), ((2, 3), 128, 48, "bf16", False), (9, 70, 33, "f16", True), (17, 256, 64, "bf16", True)]):
        lin = torch.nn.Linear(K, N, bias=has_bias)
        with torch.no_grad():
            lin.weight.copy_(synthetic.normal((N, K), torch.float32, seed=920 + 3 * ci, std=0.05))
            if has_bias:
                lin.bias.copy_(synthetic.normal((N,), torch.float32, seed=921 + 3 * ci))
        lin = lin.to(DT[dt])
        l8 = LinearFP8.from_linear(lin)
        lead = M if isinstance(M, tuple) else (M,)
        x = synthetic.normal(lead + (K,), DT[dt], seed=922 + 3 * ci)
        y = l8(x)
        arrays[f"l{ci}_W"] = bits(lin.weight.data)
        if has_bias:
            arrays[f"l{ci}_bias"] = bits(lin.bias.data)
        arrays[f"l{ci}_x"], arrays[f"l{ci}_y"] = bits(x), bits(y)
        arrays[f"l{ci}_q"], arrays[f"l{ci}_s"] = bits(l8.weight_fp8), bits(l8.weight_scales)
        cases.append(dict(kind="linear_fp8", id=ci, M=list(lead), K=K, N=N, dtype=dt, bias=has_bias,
                          state_keys=sorted(l8.state_dict().keys())))
    np.savez_compressed(os.path.join(HERE, "g7_fp8.npz"), **arrays)
    with open(os.path.join(HERE, "manifest_fp8.json"), "w") as f:
        json.dump(dict(provenance=dict(reference="mpsops/mps-bitsandbytes v%s (/root/reference, CPU path)" % ref.__version__,
                                       torch=torch.__version__, generated=time.strftime("%Y-%m-%d"),
                                       script="tests/golden/make_golden_fp8.py"), g7=cases), f, indent=1)
    print("wrote g7_fp8.npz", len(arrays), "arrays")


if __name__ == "__main__":
    main()
