"""CPU oracle vs the reference's outputs for the SURVEY §8f rank-3 rows (Embedding4bit / Embedding8bit /
OutlierAwareLinear): tests/golden/g6_nn.npz was captured from the reference by make_golden_nn.py."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from tests.goldenio import DT, HERE, bits_equal, from_bits, rel_fro


@pytest.fixture(scope="module")
def g6():
    with open(os.path.join(HERE, "manifest_nn.json")) as f:
        cases = json.load(f)["g6"]
    return cases, np.load(os.path.join(HERE, "g6_nn.npz"))


def _cases(g6, kind):
    return [c for c in g6[0] if c["kind"] == kind]


def test_embedding4bit_quantize_and_lookup_bit_exact(g6):
    z = g6[1]
    for c in _cases(g6, "embedding4bit"):
        i, dt = c["id"], DT[c["dtype"]]
        W = from_bits(z[f"e4{i}_W"], dt)
        # from_embedding quantises row by row (nn/embedding.py:183-193) == the 2-D row-wise quantiser when dim % blocksize == 0
        packed, absmax, _ = oracle.quantize_4bit(W, blocksize=c["blocksize"], quant_type=c["quant_type"])
        assert bits_equal(packed.reshape(c["num"], c["dim"] // 2), from_bits(z[f"e4{i}_packed"]))
        assert bits_equal(absmax.reshape(c["num"], -1), from_bits(z[f"e4{i}_absmax"]))
        y = oracle.embedding_4bit(torch.from_numpy(z[f"e4{i}_idx"]), from_bits(z[f"e4{i}_packed"]), from_bits(z[f"e4{i}_absmax"]),
                                  c["dim"], c["blocksize"], c["quant_type"], c["padding_idx"], dt)
        assert bits_equal(y, from_bits(z[f"e4{i}_y"], dt)), c


def test_embedding8bit_quantize_and_lookup_bit_exact(g6):
    z = g6[1]
    for c in _cases(g6, "embedding8bit"):
        i, dt = c["id"], DT[c["dtype"]]
        q, s = oracle.quantize_rowwise(from_bits(z[f"e8{i}_W"], dt))
        assert bits_equal(q, from_bits(z[f"e8{i}_q"])) and bits_equal(s, from_bits(z[f"e8{i}_s"]))
        y = oracle.embedding_8bit(torch.from_numpy(z[f"e8{i}_idx"]), q, s, c["padding_idx"], dt)
        assert bits_equal(y, from_bits(z[f"e8{i}_y"], dt)), c


def test_outlier_linear_matches_reference(g6):
    z = g6[1]
    for c in _cases(g6, "outlier_linear"):
        i, dt = c["id"], DT[c["dtype"]]
        W = from_bits(z[f"oa{i}_W"], dt)
        # from_linear (nn/outlier_aware.py:183-206): outlier columns, zeroed before the row-wise quantiser
        col_max = W.abs().max(dim=0).values
        oidx = torch.where(col_max > c["threshold"] * W.abs().mean())[0]
        assert torch.equal(oidx, torch.from_numpy(z[f"oa{i}_oidx"])) and sorted(oidx.tolist()) == sorted(c["planted"])
        W0 = W.clone()
        W0[:, oidx] = 0
        q, s = oracle.quantize_rowwise(W0)
        assert bits_equal(q, from_bits(z[f"oa{i}_q"])) and bits_equal(s, from_bits(z[f"oa{i}_s"]))
        assert bits_equal(W[:, oidx].contiguous(), from_bits(z[f"oa{i}_ow"], dt).reshape(c["N"], -1))
        bias = from_bits(z[f"oa{i}_bias"], dt) if c["bias"] else None
        x = from_bits(z[f"oa{i}_x"], dt).reshape(*c["M"], c["K"])
        y = oracle.outlier_linear(x, q, s, oidx, W[:, oidx], bias)
        ref = from_bits(z[f"oa{i}_y"], dt).reshape(*c["M"], c["N"])
        assert y.shape == ref.shape
        assert rel_fro(y, ref) <= (2e-4 if dt == torch.float16 else 2e-3), (c, rel_fro(y, ref))
