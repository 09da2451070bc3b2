"""
GPU parity for the SURVEY §8f rank-3 rows: Embedding4bit / Embedding8bit / OutlierAwareLinear through the C ABI
(mbnb_embedding_4bit, mbnb_embedding_8bit, mbnb_outlier_linear) against (a) the reference's outputs
(tests/golden/g6_nn.npz) and (b) the CPU oracle on larger seeded inputs.  Embeddings: bit-exact.
OutlierAwareLinear: Frobenius rel-err; the kernel contracts exact integers on the int8 MFMA where the reference
multiplies dtype-rounded dequantised operands, so its gate is 1e-3 (fp16) / 4e-3 (bf16), hard tolerance 1e-2.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle
import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import _native, synthetic
from tests.goldenio import DT, HERE, bits_equal, from_bits, n_mismatch, rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"
OA_TOL = {torch.float16: 1e-3, torch.bfloat16: 4e-3}


@pytest.fixture(scope="module")
def g6():
    with open(os.path.join(HERE, "manifest_nn.json")) as f:
        cases = json.load(f)["g6"]
    return cases, np.load(os.path.join(HERE, "g6_nn.npz"))


def _emb(num, dim, dt, pad, seed):
    e = torch.nn.Embedding(num, dim, padding_idx=pad)
    with torch.no_grad():
        e.weight.copy_(synthetic.normal((num, dim), torch.float32, seed=seed, std=0.5))
    return e.to(dt)


def test_embedding4bit_golden_bit_exact(g6):
    cases, z = g6
    for c in [c for c in cases if c["kind"] == "embedding4bit"]:
        i, dt = c["id"], DT[c["dtype"]]
        emb = torch.nn.Embedding(c["num"], c["dim"], padding_idx=c["padding_idx"]).to(dt)
        emb.weight.data.copy_(from_bits(z[f"e4{i}_W"], dt))
        e4 = bnb.Embedding4bit.from_embedding(emb.to(DEV), quant_type=c["quant_type"], blocksize=c["blocksize"])
        assert sorted(e4.state_dict().keys()) == c["state_keys"]
        assert bits_equal(e4.weight_packed.cpu(), from_bits(z[f"e4{i}_packed"]))
        assert bits_equal(e4.weight_absmax.cpu(), from_bits(z[f"e4{i}_absmax"]))
        idx = torch.from_numpy(z[f"e4{i}_idx"])
        y = e4(idx.to(DEV))
        assert y.dtype == dt and tuple(y.shape) == tuple(idx.shape) + (c["dim"],)
        assert n_mismatch(y.cpu(), from_bits(z[f"e4{i}_y"], dt).reshape(y.shape)) == 0, c
        assert _native.last_kernel() == "embedding4"
        if c["padding_idx"] is not None:
            assert float(y[0, 0].abs().max()) == 0.0


def test_embedding8bit_golden_bit_exact(g6):
    cases, z = g6
    for c in [c for c in cases if c["kind"] == "embedding8bit"]:
        i, dt = c["id"], DT[c["dtype"]]
        emb = torch.nn.Embedding(c["num"], c["dim"], padding_idx=c["padding_idx"]).to(dt)
        emb.weight.data.copy_(from_bits(z[f"e8{i}_W"], dt))
        e8 = bnb.Embedding8bit.from_embedding(emb.to(DEV))
        assert sorted(e8.state_dict().keys()) == c["state_keys"]
        assert bits_equal(e8.weight_int8.cpu(), from_bits(z[f"e8{i}_q"])) and bits_equal(e8.weight_scales.cpu(), from_bits(z[f"e8{i}_s"]))
        idx = torch.from_numpy(z[f"e8{i}_idx"])
        y = e8(idx.to(DEV))
        assert n_mismatch(y.cpu(), from_bits(z[f"e8{i}_y"], dt).reshape(y.shape)) == 0, c
        assert _native.last_kernel() == "embedding8"


@pytest.mark.parametrize("num,dim,qt,bs,dt,pad", [(32000, 4096, "nf4", 64, torch.bfloat16, 0), (5000, 1024, "fp4", 128, torch.float16, None),
                                                   (300, 96, "nf4", 32, torch.float16, 7), (128, 4096, "nf4", 64, torch.float32, None)])
def test_embedding4bit_vs_oracle(num, dim, qt, bs, dt, pad):
    """LLM-sized table (32000 x 4096), ragged dims (96: scalar tail path), f32 output; 4096 looked-up rows with repeats."""
    src = dt if dt != torch.float32 else torch.float16
    W = synthetic.normal((num, dim), src, seed=801, std=0.5)
    packed, absmax, _ = oracle.quantize_4bit(W, blocksize=bs, quant_type=qt)
    packed, absmax = packed.reshape(num, dim // 2), absmax.reshape(num, -1)
    idx = torch.from_numpy((synthetic.uniform_u64(4096, seed=802) % np.uint64(num)).astype(np.int64)).reshape(8, 512)
    if pad is not None:
        idx[3, 17] = pad
    y = bnb.embedding_4bit(idx.to(DEV), packed.to(DEV), absmax.to(DEV), dim, bs, qt, pad, dt)
    ref = oracle.embedding_4bit(idx, packed, absmax, dim, bs, qt, pad, dt)
    assert n_mismatch(y.cpu(), ref) == 0


@pytest.mark.parametrize("num,dim,dt,pad", [(32000, 4096, torch.bfloat16, 1), (777, 70, torch.float16, None)])
def test_embedding8bit_vs_oracle(num, dim, dt, pad):
    W = synthetic.normal((num, dim), dt, seed=811, std=0.5)
    q, s = oracle.quantize_rowwise(W)
    idx = torch.from_numpy((synthetic.uniform_u64(2048, seed=812) % np.uint64(num)).astype(np.int64))
    if pad is not None:
        idx[5] = pad
    y = bnb.embedding_8bit(idx.to(DEV), q.to(DEV), s.to(DEV), pad, dt)
    assert n_mismatch(y.cpu(), oracle.embedding_8bit(idx, q, s, pad, dt)) == 0


def test_embedding_out_of_range_rows_are_zero_and_module_validation():
    e = bnb.Embedding8bit(10, 16, device=DEV)
    y = e(torch.tensor([3, 10, -1], device=DEV))
    assert float(y[1].abs().max()) == 0.0 and float(y[2].abs().max()) == 0.0
    with pytest.raises(ValueError, match="embedding_dim must be even"):
        bnb.Embedding4bit(10, 15)
    with pytest.raises(ValueError, match="quant_type must be"):
        bnb.Embedding4bit(10, 16, quant_type="int4")
    assert isinstance(bnb.EmbeddingNF4(100, 64, device=DEV), bnb.Embedding4bit) and bnb.EmbeddingFP4(100, 64).quant_type == "fp4"


def test_outlier_linear_golden(g6):
    cases, z = g6
    for c in [c for c in cases if c["kind"] == "outlier_linear"]:
        i, dt = c["id"], DT[c["dtype"]]
        lin = torch.nn.Linear(c["K"], c["N"], bias=c["bias"]).to(dt)
        lin.weight.data.copy_(from_bits(z[f"oa{i}_W"], dt))
        if c["bias"]:
            lin.bias.data.copy_(from_bits(z[f"oa{i}_bias"], dt))
        oa = bnb.OutlierAwareLinear.from_linear(lin.to(DEV), threshold=c["threshold"])
        assert sorted(oa.state_dict().keys()) == c["state_keys"]
        assert torch.equal(oa.outlier_indices.cpu(), torch.from_numpy(z[f"oa{i}_oidx"]))
        assert bits_equal(oa.weight_int8.cpu(), from_bits(z[f"oa{i}_q"])) and bits_equal(oa.weight_scales.cpu(), from_bits(z[f"oa{i}_s"]))
        assert bits_equal(oa.outlier_weights.cpu().contiguous(), from_bits(z[f"oa{i}_ow"], dt).reshape(c["N"], -1))
        x = from_bits(z[f"oa{i}_x"], dt).reshape(*c["M"], c["K"])
        y = oa(x.to(DEV))
        ref = from_bits(z[f"oa{i}_y"], dt).reshape(*c["M"], c["N"])
        assert y.shape == ref.shape and y.dtype == dt
        assert rel_fro(y.cpu(), ref) <= OA_TOL[dt], (c, rel_fro(y.cpu(), ref))


@pytest.mark.parametrize("M,K,N,dt,n_out,bias", [(512, 4096, 4096, torch.float16, 12, True), (300, 1024, 777, torch.bfloat16, 0, True),
                                                  (64, 200, 96, torch.float16, 3, False), (2048, 4096, 4096, torch.bfloat16, 40, False),
                                                  (2560, 512, 2560, torch.float16, 21, True), (2600, 256, 2500, torch.bfloat16, 64, True),
                                                  (4096, 4096, 4096, torch.bfloat16, 16, True), (2560, 512, 2560, torch.bfloat16, 0, True),
                                                  (2500, 384, 2608, torch.float16, 32, False), (2560, 640, 2568, torch.bfloat16, 7, True),
                                                  (2560, 256, 2508, torch.bfloat16, 5, True), (2400, 512, 2507, torch.float16, 16, True),
                                                  (2560, 512, 2560, torch.float16, 70, True), (2304, 1024, 2816, torch.float16, 33, False)])
def test_outlier_linear_vs_oracle(M, K, N, dt, n_out, bias):
    """MFMA-sized shapes (256^2 and 128^2 int8 kernels), ragged K (generic kernel), with / without outliers and bias; on the
    eight-wave 256^2 kernel the outlier columns ride in the epilogue in chunks of 16 (40 and 64 columns: three / four chunks,
    ragged last chunk; here 70); with at most 64 of them on >= 96 tiles the four-wave kernel of gemm_dense.h carries them in one
    or two chunks of 32 (33, 40, 64: two; 21: weight
    rows not 16-byte aligned; 0 + bias; ragged M / N, N not a multiple of 8 / odd: scalar stores and the bias tail; the 4096^3 bench
    shape)."""
    want = None
    if K % 128 == 0 and K >= 256 and ((M + 255) // 256) * ((N + 255) // 256) >= 96 and (n_out > 0 or bias):
        want = "i8_dense+outliers" if n_out <= 64 else "i8_mfma256"
    W = synthetic.normal((N, K), torch.float32, seed=821, std=0.05)
    oidx = torch.from_numpy(np.sort((synthetic.uniform_u64(4 * n_out + 1, seed=822) % np.uint64(K)).astype(np.int64))).unique()[:n_out]
    W[:, oidx] *= 30.0
    W = W.to(dt)
    W0 = W.clone()
    W0[:, oidx] = 0
    q, s = oracle.quantize_rowwise(W0)
    ow = W[:, oidx].contiguous()
    b = synthetic.normal((N,), dt, seed=823) if bias else None
    x = synthetic.normal((M, K), dt, seed=824)
    y = bnb.outlier_linear(x.to(DEV), q.to(DEV), s.to(DEV), oidx.to(DEV), ow.to(DEV), None if b is None else b.to(DEV), dt)
    if want is not None:
        assert _native.last_kernel() == want, _native.last_kernel()
    rows = torch.cat([torch.arange(0, M, max(1, M // 60))[:60], torch.arange(M - 4, M)])   # incl. the ragged last tile           # rows are independent: a row sample bounds the oracle's time
    ref = oracle.outlier_linear(x[rows], q, s, oidx, ow, b)
    err = rel_fro(y.cpu()[rows], ref)
    assert err <= OA_TOL[dt], err


# --------------------------------------------------------------------------- FP8 E4M3
@pytest.fixture(scope="module")
def g7():
    with open(os.path.join(HERE, "manifest_fp8.json")) as f:
        cases = json.load(f)["g7"]
    return cases, np.load(os.path.join(HERE, "g7_fp8.npz"))


def test_fp8_quantize_dequantize_golden_bit_exact(g7):
    """The reference's encoder incl. its exponent rule just below powers of two, clamps, zeros, and its decoder."""
    cases, z = g7
    for c in [c for c in cases if c["kind"] == "quant"]:
        n = c["name"]
        x = from_bits(z[f"q_{n}_x"], DT[c["dtype"]]).reshape(c["shape"])
        q, s = bnb.quantize_fp8_e4m3(x.to(DEV))
        assert n_mismatch(q.cpu(), from_bits(z[f"q_{n}_q"]).reshape(c["shape"])) == 0, n
        assert bits_equal(s.cpu(), from_bits(z[f"q_{n}_s"])), n
        for dt in ("f16", "bf16", "f32"):
            deq = bnb.dequantize_fp8_e4m3(q, s, DT[dt])
            assert n_mismatch(deq.cpu(), from_bits(z[f"q_{n}_deq_{dt}"], DT[dt]).reshape(c["shape"])) == 0, (n, dt)
    allb = torch.arange(256, dtype=torch.uint8).reshape(2, 128)
    got = bnb.dequantize_fp8_e4m3(allb.to(DEV), torch.tensor([1.0, 0.37], device=DEV), torch.float32).cpu()
    ref = from_bits(z["dec_all"]).reshape(2, 128)
    assert bool(((got.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(got) & torch.isnan(ref))).all())


def test_fp8_quantize_full_size_vs_oracle():
    for dt, std in ((torch.float16, 1.0), (torch.bfloat16, 0.02), (torch.float32, 30.0)):
        x = synthetic.normal((2048, 4096), dt, seed=931, std=std)
        q, s = bnb.quantize_fp8_e4m3(x.to(DEV))
        oq, os_ = oracle.quantize_fp8_e4m3(x)
        assert n_mismatch(q.cpu(), oq) == 0 and bits_equal(s.cpu(), os_)
        assert n_mismatch(bnb.dequantize_fp8_e4m3(q, s, dt).cpu(), oracle.dequantize_fp8_e4m3(oq, os_, dt)) == 0


@pytest.mark.parametrize("dt,emin,emax", [(torch.float32, -110, 110), (torch.bfloat16, -110, 110), (torch.float16, -14, 11)])
def test_fp8_quantize_dynamic_range_bit_exact(dt, emin, emax):
    """v / scale is a true division in the reference (functional.py:1086-1104) and the encoder reads the exponent off log2.
    Rows scaled by 2^e over the dtype's range, numerators below 2^-90 of both signs, signed zeros and exact powers of two
    planted inside normal rows: bytes and scales equal the oracle's.  Pins the integer form of the encoder (exponent / mantissa
    fields instead of log2 / division: common.h float_to_fp8_e4m3) and the shared-reciprocal division of the kernel."""
    rows, cols = 663, 1024
    x = synthetic.normal((rows, cols), torch.float32, seed=941)
    e = torch.arange(rows) % (emax - emin + 1) + emin
    x = x * torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double()).float().unsqueeze(1)
    x[:, 5] = 0.0
    x[:, 6] = -0.0
    x[:, 7] = torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double()).float()            # exact powers of two
    if dt != torch.float16:
        x[:, 8] = -(2.0 ** -100)
        x[:, 9] = 2.0 ** -120
        x[:, 10] = -(2.0 ** -131)
    x = x.to(dt)
    assert torch.isfinite(x.float()).all()
    q, s = bnb.quantize_fp8_e4m3(x.to(DEV))
    oq, os_ = oracle.quantize_fp8_e4m3(x)
    assert bits_equal(s.cpu(), os_)
    assert n_mismatch(q.cpu(), oq) == 0


def test_linear_fp8_golden(g7):
    cases, z = g7
    for c in [c for c in cases if c["kind"] == "linear_fp8"]:
        i, dt = c["id"], DT[c["dtype"]]
        lin = torch.nn.Linear(c["K"], c["N"], bias=c["bias"]).to(dt)
        lin.weight.data.copy_(from_bits(z[f"l{i}_W"], dt).reshape(c["N"], c["K"]))
        if c["bias"]:
            lin.bias.data.copy_(from_bits(z[f"l{i}_bias"], dt))
        l8 = bnb.LinearFP8.from_linear(lin.to(DEV))
        assert sorted(l8.state_dict().keys()) == c["state_keys"]
        assert n_mismatch(l8.weight_fp8.cpu(), from_bits(z[f"l{i}_q"]).reshape(c["N"], c["K"])) == 0
        assert bits_equal(l8.weight_scales.cpu(), from_bits(z[f"l{i}_s"]))
        x = from_bits(z[f"l{i}_x"], dt).reshape(*c["M"], c["K"])
        y = l8(x.to(DEV))
        ref = from_bits(z[f"l{i}_y"], dt).reshape(*c["M"], c["N"])
        assert y.shape == ref.shape and y.dtype == dt
        assert rel_fro(y.cpu(), ref) <= (2e-4 if dt == torch.float16 else 2e-3), (c, rel_fro(y.cpu(), ref))


@pytest.mark.parametrize("M,N,K,dt,kern", [(1, 4096, 4096, torch.bfloat16, "fp8a16_skinny"), (33, 1000, 384, torch.float16, "fp8a16_skinny"),
                                            (300, 1000, 1024, torch.float16, "fp8a16_mfma128_splitk"), (200, 384, 80, torch.bfloat16, "fp8a16_mfma128"),
                                            (150, 1000, 1024, torch.bfloat16, "fp8a16_small_splitk"), (48, 4096, 512, torch.float16, "fp8a16_small"),
                                            (2560, 2560, 256, torch.bfloat16, "fp8a16_dequant+dense"), (3, 50, 20, torch.float16, "fp8a16_generic")])
def test_linear_fp8_kernels_vs_oracle(M, N, K, dt, kern):
    W = synthetic.normal((N, K), dt, seed=941, std=0.05)
    q, s = oracle.quantize_fp8_e4m3(W)
    x = synthetic.normal((M, K), dt, seed=942)
    b = synthetic.normal((N,), dt, seed=943)
    y = bnb.matmul_fp8_e4m3(x.to(DEV), q.to(DEV), s.to(DEV), b.to(DEV), dt)
    assert _native.last_kernel() == kern
    rows = torch.arange(0, M, max(1, M // 64))[:64]
    ref = oracle.linear_fp8(x[rows], q, s, b)
    assert rel_fro(y.cpu()[rows], ref) <= (2e-4 if dt == torch.float16 else 2e-3)


def test_fp8_hardware_decoder_matches_reference_decoder_for_every_byte():
    """The GEMM producers decode FP8 with gfx950's v_cvt_f32_fp8; the dequantize kernel uses the portable restatement of
    the reference's decoder.  One-hot activations pull every byte value (subnormals included) through the GEMM path."""
    codes = torch.arange(256, dtype=torch.int64)
    codes[0x7F], codes[0xFF] = 0, 0x80                       # NaN bytes: checked separately below
    W = codes.to(torch.uint8).reshape(2, 128).to(DEV)
    scales = torch.tensor([1.0, 1.0], device=DEV)
    want = bnb.dequantize_fp8_e4m3(W, scales, torch.float32)                      # [2, 128], portable decoder
    for half in range(2):
        x = torch.zeros(64, 128, dtype=torch.float16, device=DEV)
        x[torch.arange(64), 64 * half + torch.arange(64)] = 1.0
        y = bnb.matmul_fp8_e4m3(x, W, scales, None, torch.float16)               # [64, 2]
        assert _native.last_kernel() == "fp8a16_skinny"
        ref = want[:, 64 * half:64 * half + 64].t().to(torch.float16)
        assert torch.equal(y.float(), ref.float()), half                          # -0.0 == +0.0
    Wn = torch.full((1, 128), 0x38, dtype=torch.uint8, device=DEV)                # 1.0 everywhere ...
    Wn[0, 5] = 0x7F                                                               # ... and one NaN byte
    y = bnb.matmul_fp8_e4m3(torch.ones(2, 128, dtype=torch.float16, device=DEV), Wn, torch.ones(1, device=DEV), None, torch.float16)
    assert bool(torch.isnan(y).all())
