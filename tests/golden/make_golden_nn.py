#!/usr/bin/env python3
"""
Golden vectors for the SURVEY §8f rank-3 rows — Embedding4bit / Embedding8bit / OutlierAwareLinear — captured by
RUNNING THE REFERENCE's Python CPU path in the build container (same rules as make_golden.py: data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nn.py

Writes g6_nn.npz (bit patterns) and manifest_nn.json (case list) next to this file.
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

import mps_bitsandbytes as ref  # noqa: E402  (the reference, CPU path)
from mps_bitsandbytes.nn import Embedding4bit, Embedding8bit, OutlierAwareLinear  # noqa: E402
from mps_bitsandbytes_amd import synthetic  # noqa: E402
from make_golden import bits, DT  # noqa: E402


def main():
    arrays, cases = {}, []
    # ---- Embedding4bit (nn/embedding.py:20-199)
    specs = [(50, 64, "nf4", 64, "f16", None), (37, 128, "fp4", 32, "bf16", 3), (20, 256, "nf4", 128, "f16", 0),
             (16, 64, "nf4", 16, "bf16", None)]
    for ci, (num, dim, qt, bs, dt, pad) in enumerate(specs):
        emb = torch.nn.Embedding(num, dim, padding_idx=pad)
        with torch.no_grad():
            emb.weight.copy_(synthetic.normal((num, dim), torch.float32, seed=700 + ci, std=0.5))
        emb = emb.to(DT[dt])
        e4 = Embedding4bit.from_embedding(emb, quant_type=qt, blocksize=bs)
        idx = torch.from_numpy((synthetic.uniform_u64(21, seed=710 + ci) % np.uint64(num)).astype(np.int64)).reshape(3, 7)
        if pad is not None:
            idx[0, 0] = pad
            idx[2, 5] = pad
        y = e4(idx)
        arrays[f"e4{ci}_W"] = bits(emb.weight.data)
        arrays[f"e4{ci}_packed"], arrays[f"e4{ci}_absmax"] = bits(e4.weight_packed), bits(e4.weight_absmax)
        arrays[f"e4{ci}_idx"], arrays[f"e4{ci}_y"] = idx.numpy(), bits(y)
        cases.append(dict(kind="embedding4bit", id=ci, num=num, dim=dim, quant_type=qt, blocksize=bs, dtype=dt,
                          padding_idx=pad, state_keys=sorted(e4.state_dict().keys())))
    # ---- Embedding8bit (nn/embedding.py:202-303)
    for ci, (num, dim, dt, pad) in enumerate([(50, 64, "f16", None), (33, 70, "bf16", 2), (10, 256, "f16", 9)]):
        emb = torch.nn.Embedding(num, dim, padding_idx=pad)
        with torch.no_grad():
            emb.weight.copy_(synthetic.normal((num, dim), torch.float32, seed=730 + ci, std=0.5))
        emb = emb.to(DT[dt])
        e8 = Embedding8bit.from_embedding(emb)
        idx = torch.from_numpy((synthetic.uniform_u64(18, seed=740 + ci) % np.uint64(num)).astype(np.int64)).reshape(2, 9)
        if pad is not None:
            idx[1, 1] = pad
        y = e8(idx)
        arrays[f"e8{ci}_W"] = bits(emb.weight.data)
        arrays[f"e8{ci}_q"], arrays[f"e8{ci}_s"] = bits(e8.weight_int8), bits(e8.weight_scales)
        arrays[f"e8{ci}_idx"], arrays[f"e8{ci}_y"] = idx.numpy(), bits(y)
        cases.append(dict(kind="embedding8bit", id=ci, num=num, dim=dim, dtype=dt, padding_idx=pad,
                          state_keys=sorted(e8.state_dict().keys())))
    # ---- OutlierAwareLinear (nn/outlier_aware.py:18-219): planted outlier columns, with / without bias / outliers
    specs = [(8, 128, 64, "f16", True, 6.0, (5, 77)), ((2, 3), 96, 40, "bf16", False, 6.0, (0, 50, 95)),
             (16, 256, 128, "f16", True, 1e9, ()), (5, 70, 33, "f16", True, 4.0, (69,))]
    for ci, (M, K, N, dt, has_bias, thr, outl) in enumerate(specs):
        lin = torch.nn.Linear(K, N, bias=has_bias)
        with torch.no_grad():
            W = synthetic.normal((N, K), torch.float32, seed=760 + 3 * ci, std=0.05)
            for c in outl:
                W[:, c] *= 40.0
            lin.weight.copy_(W)
            if has_bias:
                lin.bias.copy_(synthetic.normal((N,), torch.float32, seed=761 + 3 * ci))
        lin = lin.to(DT[dt])
        oa = OutlierAwareLinear.from_linear(lin, threshold=thr)
        lead = M if isinstance(M, tuple) else (M,)
        x = synthetic.normal(lead + (K,), DT[dt], seed=762 + 3 * ci)
        y = oa(x)
        arrays[f"oa{ci}_W"] = bits(lin.weight.data)
        if has_bias:
            arrays[f"oa{ci}_bias"] = bits(lin.bias.data)
        arrays[f"oa{ci}_q"], arrays[f"oa{ci}_s"] = bits(oa.weight_int8), bits(oa.weight_scales)
        arrays[f"oa{ci}_oidx"] = oa.outlier_indices.numpy()
        arrays[f"oa{ci}_ow"] = bits(oa.outlier_weights)
        arrays[f"oa{ci}_x"], arrays[f"oa{ci}_y"] = bits(x), bits(y)
        cases.append(dict(kind="outlier_linear", id=ci, M=list(lead), K=K, N=N, dtype=dt, bias=has_bias, threshold=thr,
                          planted=list(outl), n_outliers=int(oa.outlier_indices.numel()),
                          state_keys=sorted(oa.state_dict().keys())))
    np.savez_compressed(os.path.join(HERE, "g6_nn.npz"), **arrays)
    manifest = dict(provenance=dict(reference="mpsops/mps-bitsandbytes v%s (/root/reference, CPU path)" % ref.__version__,
                                    torch=torch.__version__, generated=time.strftime("%Y-%m-%d"),
                                    script="tests/golden/make_golden_nn.py"), g6=cases)
    with open(os.path.join(HERE, "manifest_nn.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote g6_nn.npz:", {c["kind"] + str(c["id"]): c.get("n_outliers") for c in cases})


if __name__ == "__main__":
    main()
