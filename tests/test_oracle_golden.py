"""
CPU oracle (oracle/oracle.c) vs the golden vectors captured from the reference's own
Python CPU path (tests/golden/make_golden.py).  This is what PINS the oracle:
bit-exact for quantize / pack / dequantize / int8; stated tolerance for matmuls.
No GPU needed.
"""
import hashlib

import numpy as np
import pytest
import torch

import oracle
from mps_bitsandbytes_amd import synthetic
from tests.goldenio import DT, bits_equal, from_bits, n_mismatch, rel_fro


def _x(npz, key, case):
    return from_bits(npz[key + "x"], DT[case["dtype"]]).reshape(case["shape"])


def _check_quant4(npz, key, case):
    x = _x(npz, key, case)
    packed, absmax, st2 = oracle.quantize_4bit(x, case["blocksize"], case["quant_type"],
                                               case["compress_statistics"])
    g_packed = from_bits(npz[key + "packed"])
    assert n_mismatch(packed, g_packed) == 0, f"{key}: packed bytes differ"
    if case["compress_statistics"]:
        assert bits_equal(absmax, from_bits(npz[key + "absmax"])), f"{key}: int8 absmax differs"
        assert bits_equal(st2[0], from_bits(npz[key + "absmax2"])), f"{key}: absmax2 differs"
    else:
        assert bits_equal(absmax, from_bits(npz[key + "absmax"])), f"{key}: absmax differs"
    deq = oracle.dequantize_4bit(g_packed, absmax, case["shape"], case["blocksize"], case["quant_type"],
                                 DT[case["dtype"]], st2)
    g_deq = from_bits(npz[key + "deq"], DT[case["dtype"]]).reshape(case["shape"])
    assert n_mismatch(deq, g_deq) == 0, f"{key}: dequantized bits differ"


def test_g1_quantize_dequantize_bit_exact(golden):
    npz = golden.npz("g1_quant4.npz")
    assert len(golden.manifest["g1"]) >= 40
    for case in golden.manifest["g1"]:
        _check_quant4(npz, f"c{case['id']}_", case)


def test_g2_adversarial_bit_exact(golden):
    npz = golden.npz("g2_adversarial.npz")
    for case in golden.manifest["g2"]:
        _check_quant4(npz, case["id"] + "_", case)


def test_g2_known_answers(golden):
    """all-zero block -> idx 7 (NF4) / 0 (FP4); -0.0 never maps to FP4 idx 8."""
    npz = golden.npz("g2_adversarial.npz")
    assert set(np.unique(npz["zeros_nf4_pl_packed"])) == {0x77}
    assert set(np.unique(npz["zeros_fp4_pl_packed"])) == {0x00}
    assert set(np.unique(npz["negzeros_fp4_pl_packed"])) == {0x00}


def _sha(t):
    t = t.contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        b = t.view(torch.int16).numpy().tobytes()
    elif t.dtype == torch.float32:
        b = t.view(torch.int32).numpy().tobytes()
    else:
        b = t.numpy().tobytes()
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("name", ["A", "A_fp4", "B"])
def test_g3_config_digests(golden, name):
    """BASELINE configs[0] (4096x4096 fp16 bs64 round trip on CPU) and config B (11008x4096 bf16 + DQ)."""
    g = golden.g3[name]
    x = synthetic.normal(g["shape"], DT[g["dtype"]], seed=g["seed"], std=g["std"])
    assert _sha(x) == g["input"], "synthetic PRNG does not reproduce the golden input"
    packed, absmax, st2 = oracle.quantize_4bit(x, g["blocksize"], g["quant_type"], g["compress_statistics"])
    assert packed.numel() == g["packed_numel"] and absmax.numel() == g["absmax_numel"]
    assert _sha(packed) == g["packed"]
    assert _sha(absmax) == g["absmax"]
    if st2 is not None:
        assert _sha(st2[0]) == g["absmax2"]
    deq = oracle.dequantize_4bit(packed, absmax, g["shape"], g["blocksize"], g["quant_type"], DT[g["dtype"]], st2)
    assert _sha(deq) == g["deq"]


def test_g3_rowwise_digest(golden):
    g = golden.g3["A_rowwise"]
    x = synthetic.normal(g["shape"], DT[g["dtype"]], seed=g["seed"], std=g["std"])
    q, s = oracle.quantize_rowwise(x)
    assert _sha(q) == g["q"] and _sha(s) == g["scales"]
    assert _sha(oracle.dequantize_rowwise(q, s, torch.float16)) == g["deq"]


# matmul tolerance: the reference's CPU GEMM and the oracle's differ only in f32 summation
# order, so outputs agree to a final-rounding ulp; gate on Frobenius rel-err.
MATMUL_TOL = {"f16": 2e-4, "bf16": 2e-3, "f32": 2e-6}


def test_g4_matmul_4bit(golden):
    npz = golden.npz("g4_matmul.npz")
    for c in golden.manifest["g4"]:
        key = f"c{c['id']}_"
        A = from_bits(npz[key + "A"], DT[c["a_dtype"]]).reshape(c["M"] + [c["K"]])
        packed = from_bits(npz[key + "packed"])
        if c["compress_statistics"]:
            absmax = from_bits(npz[key + "absmax"])
            st2 = (from_bits(npz[key + "absmax2"]), 256)
        else:
            absmax, st2 = from_bits(npz[key + "absmax"]), None
        bias = None if c["bias_dtype"] is None else from_bits(npz[key + "bias"], DT[c["bias_dtype"]])
        cd = None if c["compute_dtype"] is None else DT[c["compute_dtype"]]
        out = oracle.matmul_4bit(A, packed, absmax, (c["N"], c["K"]), c["blocksize"], c["quant_type"],
                                 DT[c["w_dtype"]], bias, cd, st2)
        ref = from_bits(npz[key + "out"], DT[c["out_dtype"]]).reshape(c["M"] + [c["N"]])
        assert out.dtype == ref.dtype and out.shape == ref.shape, key
        coarse = c["w_dtype"] if c["w_dtype"] != "f32" else c["out_dtype"]
        if c["out_dtype"] == "bf16":
            coarse = "bf16"
        err = rel_fro(out, ref)
        assert err <= MATMUL_TOL[coarse], f"{key}: rel-err {err:.3e} (tol {MATMUL_TOL[coarse]})"


def test_g5_rowwise(golden):
    npz = golden.npz("g5_int8.npz")
    for c in [c for c in golden.manifest["g5"] if c["kind"] == "rowwise"]:
        k = f"rw{c['id']}_"
        x = from_bits(npz[k + "x"], DT[c["dtype"]]).reshape(c["shape"])
        q, s = oracle.quantize_rowwise(x)
        assert n_mismatch(q, from_bits(npz[k + "q"])) == 0
        assert bits_equal(s, from_bits(npz[k + "s"]))
        for odt in ("f16", "bf16", "f32"):
            d = oracle.dequantize_rowwise(q, s, DT[odt])
            assert n_mismatch(d, from_bits(npz[k + "deq_" + odt], DT[odt]).reshape(c["shape"])) == 0
    for k, dt in (("rwfill_", torch.float16), ("rwtie_", torch.float32), ("rwzero_", torch.float16)):
        x = from_bits(npz[k + "x"], dt)
        q, s = oracle.quantize_rowwise(x)
        assert n_mismatch(q, from_bits(npz[k + "q"])) == 0, k
        assert bits_equal(s, from_bits(npz[k + "s"])), k
    assert set(np.unique(npz["rwfill_q"])) == {127}            # tests/test_advanced_linear.py:139-153
    assert list(npz["rwtie_q"][0][1:7]) == [0, 2, 2, 0, -2, -2]  # half-to-even


def test_g5_blockwise(golden):
    npz = golden.npz("g5_int8.npz")
    for c in [c for c in golden.manifest["g5"] if c["kind"] == "blockwise"]:
        k = f"bw{c['id']}_"
        x = from_bits(npz[k + "x"], DT[c["dtype"]])
        q, am = oracle.quantize_blockwise(x, c["blocksize"])
        assert n_mismatch(q, from_bits(npz[k + "q"])) == 0
        if c["nested"]:
            q2, am2 = oracle.quantize_blockwise(am, 256)
            assert bits_equal(q2, from_bits(npz[k + "absmax"]))
            assert bits_equal(am2, from_bits(npz[k + "absmax2"]))
            am = oracle.dequantize_blockwise(q2, am2, 256, torch.float32)
        else:
            assert bits_equal(am, from_bits(npz[k + "absmax"]))
        d = oracle.dequantize_blockwise(q, am, c["blocksize"], DT[c["dtype"]])
        assert n_mismatch(d, from_bits(npz[k + "deq"], DT[c["dtype"]])) == 0


def test_g5_matmul_int8(golden):
    npz = golden.npz("g5_int8.npz")
    for c in [c for c in golden.manifest["g5"] if c["kind"] == "matmul_int8"]:
        k = f"mm{c['id']}_"
        out = oracle.matmul_int8(from_bits(npz[k + "A"]), from_bits(npz[k + "B"]), from_bits(npz[k + "As"]),
                                 from_bits(npz[k + "Bs"]), DT[c["dtype"]])
        ref = from_bits(npz[k + "out"], DT[c["dtype"]]).reshape(c["M"], c["N"])
        assert rel_fro(out, ref) <= MATMUL_TOL[c["dtype"]], k


def test_g5_double_quant(golden):
    npz = golden.npz("g5_int8.npz")
    for c in [c for c in golden.manifest["g5"] if c["kind"] == "double_quant"]:
        k = f"dq{c['id']}_"
        x = from_bits(npz[k + "x"], DT[c["dtype"]]).reshape(c["shape"])
        oc, orow, cs, rs, outl = oracle.double_quant(x)
        assert outl is None
        assert n_mismatch(oc, from_bits(npz[k + "out_col"])) == 0
        assert n_mismatch(orow, from_bits(npz[k + "out_row"])) == 0
        assert bits_equal(cs, from_bits(npz[k + "col_stats"]))
        assert bits_equal(rs, from_bits(npz[k + "row_stats"]))


def test_g5_linear8bit(golden):
    npz = golden.npz("g5_int8.npz")
    for c in [c for c in golden.manifest["g5"] if c["kind"] == "linear8bit"]:
        k = f"l8{c['id']}_"
        dt = DT[c["dtype"]]
        W = from_bits(npz[k + "W"], dt).reshape(c["N"], c["K"])
        q, s = oracle.quantize_rowwise(W)
        assert n_mismatch(q, from_bits(npz[k + "q"])) == 0 and bits_equal(s, from_bits(npz[k + "s"]))
        x = from_bits(npz[k + "x"], dt).reshape(c["M"] + [c["K"]])
        bias = from_bits(npz[k + "bias"], dt) if c["bias"] else None
        y = oracle.linear_int8(x, q, s, bias)
        ref = from_bits(npz[k + "y"], dt).reshape(c["M"] + [c["N"]])
        assert rel_fro(y, ref) <= MATMUL_TOL[c["dtype"]], k
