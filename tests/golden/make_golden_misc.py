#!/usr/bin/env python3
"""
Golden vectors for the small round-2 rows, captured by RUNNING THE REFERENCE's Python CPU path here (data only):
  * dequant_absmax, legacy (non-QuantState) form (functional.py:866-889): 2-D / 1-D codes, int8 / uint8 / float codes,
    a ragged last scale block, and fewer scale blocks than the codes need (the zeros_like tail);
  * BitsAndBytesConfig.from_dict's string parse of the compute dtype (integration.py:79-94).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_misc.py     -> g8_misc.npz, manifest_misc.json
"""
import json
import os
import sys
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
from mps_bitsandbytes.integration import BitsAndBytesConfig  # noqa: E402
from mps_bitsandbytes_amd import synthetic  # noqa: E402


def main():
    arrays, manifest = {}, {"dequant_absmax": [], "config_from_dict": [], "quant4_nan": [], "quant4_absmax_in": []}
    # NaN inside a quantisation block: the reference's abs().max() propagates it into absmax and argmin over
    # all-NaN distances returns index 0 for the whole block; neighbouring blocks are untouched
    for i, (dt, qt, bs, shape) in enumerate([(torch.float16, "nf4", 64, (4, 256)), (torch.float32, "fp4", 64, (3, 192)),
                                             (torch.bfloat16, "nf4", 32, (2, 128)), (torch.float16, "nf4", 1024, (1, 2048))]):
        x = synthetic.normal(shape, dt, seed=400 + i)
        xf = x.view(-1)
        for pos in (5, (shape[1] + 70) % xf.numel(), xf.numel() - 3):
            xf[pos] = float("nan")
        packed, st = ref.functional.quantize_4bit(x.clone(), blocksize=bs, quant_type=qt)
        k = f"qn{i}_"
        arrays[k + "x"] = x.view(torch.int16).numpy().view(np.uint16) if dt != torch.float32 else x.numpy().view(np.uint32)
        arrays[k + "packed"] = packed.numpy()
        arrays[k + "absmax"] = st.absmax.float().contiguous().numpy().view(np.uint32)
        manifest["quant4_nan"].append({"id": i, "dtype": {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[dt],
                                       "quant_type": qt, "blocksize": bs, "shape": list(shape)})
    # caller-supplied absmax (functional.py:231: used as the normaliser when given) far below |x|: quotients of 2^0 .. 2^40 and
    # inf.  From |xn| ~ 2^23 the f32 distances to the codes tie and argmin returns the FIRST tied index, not the nearest code
    for i, (dt, qt, bs, shape, am) in enumerate([(torch.float32, "nf4", 64, (41, 128), 2.0 ** -10), (torch.float32, "fp4", 64, (41, 128), 2.0 ** -10),
                                                 (torch.bfloat16, "nf4", 32, (41, 64), 2.0 ** -30), (torch.float16, "fp4", 4, (16, 64), 2.0 ** -24)]):
        x = synthetic.normal(shape, torch.float32, seed=500 + i)
        e = torch.arange(shape[0]).double() - (10 if dt != torch.float16 else 14)
        x = (x * torch.pow(torch.tensor(2.0, dtype=torch.float64), e).float().unsqueeze(1)).to(dt)
        if dt == torch.float32:
            x[-1, :8] = torch.tensor([float("inf"), float("-inf"), 3e38, -3e38, 2.0 ** 13, -(2.0 ** 13), 2.0 ** 15 + 1, 0.0])
        nblocks = x.numel() // bs
        absmax = torch.full((shape[0], shape[1] // bs), am, dtype=torch.float32)
        packed, st = ref.functional.quantize_4bit(x.clone(), absmax=absmax.clone(), blocksize=bs, quant_type=qt)
        k = f"qa{i}_"
        arrays[k + "x"] = x.view(torch.int16).numpy().view(np.uint16) if dt != torch.float32 else x.numpy().view(np.uint32)
        arrays[k + "packed"] = packed.numpy()
        manifest["quant4_absmax_in"].append({"id": i, "dtype": {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[dt],
                                             "quant_type": qt, "blocksize": bs, "shape": list(shape), "absmax": am, "nblocks": nblocks})
    cases = [
        dict(rows=4, num_blocks=600, dq_blocks=3, blocksize=256, code="int8"),     # ragged last block (88 codes)
        dict(rows=3, num_blocks=512, dq_blocks=2, blocksize=256, code="uint8"),
        dict(rows=1, num_blocks=700, dq_blocks=3, blocksize=256, code="uint8", one_d=True),
        dict(rows=5, num_blocks=300, dq_blocks=1, blocksize=256, code="int8"),     # zeros_like tail: codes 256..299 uncovered
        dict(rows=2, num_blocks=96, dq_blocks=3, blocksize=32, code="f32"),
        dict(rows=2, num_blocks=64, dq_blocks=4, blocksize=32, code="int8"),       # more scale blocks than codes
    ]
    for i, c in enumerate(cases):
        n = c["rows"] * c["num_blocks"]
        raw = synthetic.uniform_u64(n, seed=100 + i)
        if c["code"] == "int8":
            q = torch.from_numpy((raw % np.uint64(255)).astype(np.int64) - 127).to(torch.int8)
        elif c["code"] == "uint8":
            q = torch.from_numpy((raw % np.uint64(256)).astype(np.uint8))
        else:
            q = synthetic.normal((n,), torch.float32, seed=200 + i)
        scales = synthetic.normal((c["rows"] * c["dq_blocks"],), torch.float32, seed=300 + i, std=0.01).abs() + 1e-4
        if c.get("one_d"):
            qq, ss = q.view(-1), scales.view(-1)
        else:
            qq, ss = q.view(c["rows"], c["num_blocks"]), scales.view(c["rows"], c["dq_blocks"])
        out = ref.functional.dequant_absmax(qq.clone(), ss.clone(), blocksize=c["blocksize"])
        assert out.dtype == torch.float32 and out.shape == qq.shape
        k = f"da{i}_"
        arrays[k + "q"] = q.numpy() if c["code"] != "f32" else q.numpy().view(np.uint32)
        arrays[k + "scales"] = scales.numpy().view(np.uint32)
        arrays[k + "out"] = out.contiguous().view(-1).numpy().view(np.uint32)
        manifest["dequant_absmax"].append(dict(c, id=i))
    for s_in in ("torch.float16", "torch.bfloat16", "bfloat16", "float16", "torch.float32", "fp16", ""):
        cfg = BitsAndBytesConfig.from_dict({"load_in_4bit": True, "bnb_4bit_compute_dtype": s_in})
        manifest["config_from_dict"].append({"in": s_in, "out": str(cfg.bnb_4bit_compute_dtype)})
    np.savez_compressed(os.path.join(HERE, "g8_misc.npz"), **arrays)
    with open(os.path.join(HERE, "manifest_misc.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote g8_misc.npz:", len(arrays), "arrays;", manifest["config_from_dict"])


if __name__ == "__main__":
    main()
