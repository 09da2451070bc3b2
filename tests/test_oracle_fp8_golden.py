"""CPU oracle vs the reference's outputs for FP8 E4M3 (tests/golden/g7_fp8.npz, captured by make_golden_fp8.py):
the reference's own encoder (exponent from floor(torch.log2), no mantissa carry, >= 256 -> 240, subnormals flushed),
its decoder for all 256 bytes, matmul_fp8_e4m3 / LinearFP8.  Quantize / dequantize: bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from tests.goldenio import DT, HERE, bits_equal, from_bits, n_mismatch, rel_fro


@pytest.fixture(scope="module")
def g7():
    with open(os.path.join(HERE, "manifest_fp8.json")) as f:
        cases = json.load(f)["g7"]
    return cases, np.load(os.path.join(HERE, "g7_fp8.npz"))


def test_fp8_quantize_dequantize_bit_exact(g7):
    cases, z = g7
    for c in [c for c in cases if c["kind"] == "quant"]:
        n = c["name"]
        x = from_bits(z[f"q_{n}_x"], DT[c["dtype"]]).reshape(c["shape"])
        q, s = oracle.quantize_fp8_e4m3(x)
        assert n_mismatch(q, from_bits(z[f"q_{n}_q"]).reshape(c["shape"])) == 0, n
        assert bits_equal(s, from_bits(z[f"q_{n}_s"])), n
        for dt in ("f16", "bf16", "f32"):
            deq = oracle.dequantize_fp8_e4m3(q, s, DT[dt])
            ref = from_bits(z[f"q_{n}_deq_{dt}"], DT[dt]).reshape(c["shape"])
            assert n_mismatch(deq, ref) == 0, (n, dt)


def test_fp8_decoder_all_bytes(g7):
    z = g7[1]
    allb = torch.arange(256, dtype=torch.uint8).reshape(2, 128)
    got = oracle.dequantize_fp8_e4m3(allb, torch.tensor([1.0, 0.37]), torch.float32)
    ref = from_bits(z["dec_all"]).reshape(2, 128)
    same = (got.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(got) & torch.isnan(ref))
    assert bool(same.all())


def test_linear_fp8_matches_reference(g7):
    cases, z = g7
    for c in [c for c in cases if c["kind"] == "linear_fp8"]:
        i, dt = c["id"], DT[c["dtype"]]
        W = from_bits(z[f"l{i}_W"], dt).reshape(c["N"], c["K"])
        q, s = oracle.quantize_fp8_e4m3(W)
        assert n_mismatch(q, from_bits(z[f"l{i}_q"]).reshape(c["N"], c["K"])) == 0 and bits_equal(s, from_bits(z[f"l{i}_s"]))
        x = from_bits(z[f"l{i}_x"], dt).reshape(*c["M"], c["K"])
        b = from_bits(z[f"l{i}_bias"], dt) if c["bias"] else None
        y = oracle.linear_fp8(x, q, s, b)
        ref = from_bits(z[f"l{i}_y"], dt).reshape(*c["M"], c["N"])
        assert rel_fro(y, ref) <= (2e-4 if dt == torch.float16 else 2e-3), (c, rel_fro(y, ref))
