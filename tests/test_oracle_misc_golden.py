"""Round-2 small rows pinned on reference outputs (tests/golden/make_golden_misc.py -> g8_misc.npz / manifest_misc.json):
the oracle's dequant_absmax (legacy form) bit-exactly, and BitsAndBytesConfig.from_dict's string parse."""
import json
import os

import numpy as np
import torch

import oracle
import mps_bitsandbytes_amd as bnb
from tests.goldenio import HERE, from_bits

MAN = json.load(open(os.path.join(HERE, "manifest_misc.json")))


def load_case(npz, c):
    k = f"da{c['id']}_"
    q = torch.from_numpy(np.ascontiguousarray(npz[k + "q"])) if c["code"] != "f32" else from_bits(npz[k + "q"])
    scales, out = from_bits(npz[k + "scales"]), from_bits(npz[k + "out"])
    if c.get("one_d"):
        return q.view(-1), scales.view(-1), out.view(-1)
    return q.view(c["rows"], c["num_blocks"]), scales.view(c["rows"], c["dq_blocks"]), out.view(c["rows"], c["num_blocks"])


def test_oracle_dequant_absmax_legacy_bit_exact():
    npz = np.load(os.path.join(HERE, "g8_misc.npz"))
    for c in MAN["dequant_absmax"]:
        q, scales, want = load_case(npz, c)
        got = oracle.dequant_absmax(q, scales, c["blocksize"])
        assert got.shape == want.shape and torch.equal(got.view(torch.int32), want.view(torch.int32)), c


def test_config_from_dict_string_dtype_matches_reference():
    for rec in MAN["config_from_dict"]:
        cfg = bnb.BitsAndBytesConfig.from_dict({"load_in_4bit": True, "bnb_4bit_compute_dtype": rec["in"]})
        assert str(cfg.bnb_4bit_compute_dtype) == rec["out"], rec


def check_quant4_nan(quantize):
    """quantize(x, blocksize, quant_type) -> (packed u8, absmax f32), compared with the reference's outputs: packed bytes
    bit-exact (a block holding a NaN gets index 0 throughout), absmax NaN exactly where the reference's is."""
    from tests.goldenio import DT
    npz = np.load(os.path.join(HERE, "g8_misc.npz"))
    for c in MAN["quant4_nan"]:
        k = f"qn{c['id']}_"
        x = from_bits(npz[k + "x"], DT[c["dtype"]]).reshape(c["shape"])
        packed, absmax = quantize(x, c["blocksize"], c["quant_type"])
        g_packed, g_absmax = torch.from_numpy(np.ascontiguousarray(npz[k + "packed"])), from_bits(npz[k + "absmax"])
        nan = torch.isnan(g_absmax)
        assert int(nan.sum()) >= 2, c
        assert torch.equal(packed.cpu().view(-1), g_packed.view(-1)), c
        a = absmax.cpu().view(-1)
        assert torch.equal(torch.isnan(a), nan) and torch.equal(a[~nan].view(torch.int32), g_absmax[~nan].view(torch.int32)), c


def check_quant4_absmax_in(quantize):
    """quantize(x, absmax f32 [nblocks], blocksize, quant_type) -> packed u8, against the reference run with a caller-supplied
    absmax far below |x| (functional.py:231,236-240): quotients up to 2^40 and inf, where argmin over tied f32 distances
    returns the first tied index (13 / 14 / 0), not the nearest code."""
    from tests.goldenio import DT
    npz = np.load(os.path.join(HERE, "g8_misc.npz"))
    assert len(MAN["quant4_absmax_in"]) == 4
    for c in MAN["quant4_absmax_in"]:
        k = f"qa{c['id']}_"
        x = from_bits(npz[k + "x"], DT[c["dtype"]]).reshape(c["shape"])
        absmax = torch.full((c["nblocks"],), c["absmax"], dtype=torch.float32)
        packed = quantize(x, absmax, c["blocksize"], c["quant_type"])
        g_packed = torch.from_numpy(np.ascontiguousarray(npz[k + "packed"]))
        lo = g_packed & 15
        assert int(((lo == 13) | (lo == 14)).sum()) > 0 or c["quant_type"] == "fp4", c     # the tie region is in the data
        assert torch.equal(packed.cpu().view(-1), g_packed.view(-1)), c


def test_oracle_quantize_4bit_with_supplied_absmax_ties_like_the_reference():
    check_quant4_absmax_in(lambda x, am, bs, qt: oracle.quantize_4bit(x, bs, qt, False, absmax=am)[0])


def test_oracle_quantize_4bit_propagates_nan_like_the_reference():
    check_quant4_nan(lambda x, bs, qt: oracle.quantize_4bit(x, bs, qt)[:2])
