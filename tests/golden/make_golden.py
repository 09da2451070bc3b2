#!/usr/bin/env python3
"""
Capture golden vectors by RUNNING THE REFERENCE's Python CPU path in the build
container (importing /root/reference on CPU is permitted there).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes, next to this file:
    g1_quant4.npz   G1  quantize_4bit / dequantize_4bit, assorted shapes/dtypes/blocksizes
    g2_adversarial.npz  G2  ties at code midpoints, zeros, extremes
    g3_digests.json G3  SHA-256 of packed/absmax/dequant for BASELINE configs A and B
    g4_matmul.npz   G4  matmul_4bit outputs (small shapes, biases, dtypes)
    g5_int8.npz     G5  quantize_rowwise / blockwise / matmul_int8 / double_quant / Linear8bit
    manifest.json   case list + provenance

Only DATA (inputs, parameters, expected outputs) is written; no reference source
travels.  Inputs come from mps_bitsandbytes_amd.synthetic (integer-hash PRNG) so
that the GPU box can regenerate the big ones from (shape, dtype, seed, std).
The fixtures are consumed by tests/test_oracle_golden.py (CPU oracle vs reference)
and tests/test_gpu_parity.py (HIP vs the same vectors).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import warnings  # noqa: E402

warnings.filterwarnings("ignore")

import mps_bitsandbytes as ref  # noqa: E402  (the reference, CPU fallback path)
from mps_bitsandbytes_amd import synthetic  # noqa: E402

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def bits(t: torch.Tensor) -> np.ndarray:
    """Raw bit pattern of a tensor as a numpy integer array (bf16 has no numpy dtype)."""
    t = t.detach().contiguous().cpu()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32).copy()
    return t.numpy().copy()


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(bits(t).tobytes()).hexdigest()


def state_arrays(prefix, state, out):
    out[prefix + "absmax"] = bits(state.absmax)
    if state.state2 is not None:
        out[prefix + "absmax2"] = bits(state.state2.absmax)


# ----------------------------------------------------------------------------- G1
def make_g1(manifest):
    arrays = {}
    shapes = [(64, 128), (5, 70), (7, 13), (1, 1), (128, 127), (1000,), (3, 5, 7), (16, 256)]
    dts = ["f16", "bf16", "f32"]
    bss = [64, 32, 128]
    cases = []
    i = 0
    for si, shape in enumerate(shapes):
        for qt in ("nf4", "fp4"):
            for cs in (False, True):
                dt = dts[i % 3]
                bs = bss[(i // 2) % 3]
                i += 1
                cases.append((shape, dt, bs, qt, cs))
    # the full blocksize sweep of tests/test_edge_cases.py:216-224 on a 16x16 tensor
    for bs in (2, 16, 32, 64, 128, 256, 512, 1024):
        cases.append(((16, 16), "f16", bs, "nf4", False))
    for ci, (shape, dt, bs, qt, cs) in enumerate(cases):
        seed = 100 + ci
        std = 1.0 if ci % 3 else 0.02
        x = synthetic.normal(shape, DT[dt], seed=seed, std=std)
        packed, st = ref.quantize_4bit(x, blocksize=bs, compress_statistics=cs, quant_type=qt)
        deq = ref.dequantize_4bit(packed, st)
        key = f"c{ci}_"
        arrays[key + "x"] = bits(x)
        arrays[key + "packed"] = bits(packed)
        state_arrays(key, st, arrays)
        arrays[key + "deq"] = bits(deq)
        manifest["g1"].append(dict(id=ci, shape=list(shape), dtype=dt, blocksize=bs, quant_type=qt,
                                   compress_statistics=cs, seed=seed, std=std,
                                   absmax_dtype=str(st.absmax.dtype)))
    np.savez_compressed(os.path.join(HERE, "g1_quant4.npz"), **arrays)


# ----------------------------------------------------------------------------- G2
def make_g2(manifest):
    arrays = {}
    nf4 = ref.NF4_CODEBOOK.clone()
    fp4 = ref.FP4_CODEBOOK.clone()
    cases = {}
    # exact midpoints between adjacent sorted codes; the block also holds +1.0 so absmax == 1
    for name, code in (("nf4", nf4), ("fp4", fp4)):
        srt = torch.sort(code).values
        mids = (srt[:-1] + srt[1:]) / 2
        nxt = torch.nextafter(mids, torch.tensor(2.0))
        prv = torch.nextafter(mids, torch.tensor(-2.0))
        v = torch.cat([mids, nxt, prv, srt, torch.tensor([1.0, -1.0, 0.0, -0.0])])
        pad = (-v.numel()) % 64
        v = torch.cat([v, torch.zeros(pad)])
        v[63] = 1.0
        if v.numel() > 64:
            v[127] = 1.0
        cases[f"mid_{name}_f32"] = (v.clone().reshape(1, -1), name, 64)
        cases[f"mid_{name}_f16"] = (v.clone().half().reshape(1, -1), name, 64)
        cases[f"mid_{name}_bf16"] = (v.clone().bfloat16().reshape(1, -1), name, 64)
    z = torch.zeros(4, 128, dtype=torch.float16)
    cases["zeros_nf4"] = (z, "nf4", 64)
    cases["zeros_fp4"] = (z, "fp4", 64)
    nz = torch.zeros(2, 128, dtype=torch.float16)
    nz[:, 1::2] = -0.0
    cases["negzeros_fp4"] = (nz, "fp4", 64)
    cases["negzeros_nf4"] = (nz, "nf4", 64)
    cases["f16max_nf4"] = (torch.full((2, 64), 65504.0, dtype=torch.float16), "nf4", 64)
    cases["f16max_neg_fp4"] = (torch.full((2, 64), -65504.0, dtype=torch.float16), "fp4", 64)
    cases["tiny_f32_nf4"] = (torch.full((2, 64), 1e-38, dtype=torch.float32), "nf4", 64)
    cases["tiny_f32_fp4"] = (torch.full((2, 64), -1e-38, dtype=torch.float32), "fp4", 64)
    cases["below_clamp_f32"] = (torch.full((1, 64), 5e-9, dtype=torch.float32), "nf4", 64)
    one = torch.zeros(4, 256, dtype=torch.bfloat16)
    for r in range(4):
        for b in range(4):
            one[r, b * 64 + (7 * r + 13 * b) % 64] = (-1) ** (r + b) * (0.5 + r + 3 * b)
    cases["one_nonzero_nf4"] = (one, "nf4", 64)
    cases["one_nonzero_fp4"] = (one, "fp4", 64)
    mixed = synthetic.normal((8, 64), torch.float16, seed=77)
    mixed[0, 0] = 65504.0
    mixed[1, 5] = -65504.0
    mixed[2, :] = mixed[2, :] * 1e-4
    mixed[3, 3] = 6e-8  # fp16 subnormal
    cases["mixed_extreme"] = (mixed, "nf4", 64)
    sub = torch.full((1, 64), 6e-8, dtype=torch.float16)
    cases["f16_subnormal"] = (sub, "nf4", 64)
    for name, (x, qt, bs) in cases.items():
        for cs in (False, True):
            packed, st = ref.quantize_4bit(x, blocksize=bs, compress_statistics=cs, quant_type=qt)
            deq = ref.dequantize_4bit(packed, st)
            key = f"{name}_{'dq' if cs else 'pl'}_"
            arrays[key + "x"] = bits(x)
            arrays[key + "packed"] = bits(packed)
            state_arrays(key, st, arrays)
            arrays[key + "deq"] = bits(deq)
            manifest["g2"].append(dict(id=key[:-1], shape=list(x.shape),
                                       dtype={torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[x.dtype],
                                       blocksize=bs, quant_type=qt, compress_statistics=cs))
    np.savez_compressed(os.path.join(HERE, "g2_adversarial.npz"), **arrays)


# ----------------------------------------------------------------------------- G3
def make_g3(manifest):
    out = {}
    # config A: 4096x4096 fp16 ~N(0,1), NF4 bs 64 (BASELINE configs[0], metric shape)
    t0 = time.time()
    x = synthetic.normal((4096, 4096), torch.float16, seed=1234)
    packed, st = ref.quantize_nf4(x, blocksize=64)
    deq = ref.dequantize_nf4(packed, st)
    out["A"] = dict(shape=[4096, 4096], dtype="f16", seed=1234, std=1.0, blocksize=64, quant_type="nf4",
                    compress_statistics=False, input=sha(x), packed=sha(packed), absmax=sha(st.absmax),
                    deq=sha(deq), packed_numel=packed.numel(), absmax_numel=st.absmax.numel())
    # also fp4 on the same input
    packed, st = ref.quantize_fp4(x, blocksize=64)
    deq = ref.dequantize_fp4(packed, st)
    out["A_fp4"] = dict(shape=[4096, 4096], dtype="f16", seed=1234, std=1.0, blocksize=64, quant_type="fp4",
                        compress_statistics=False, input=sha(x), packed=sha(packed), absmax=sha(st.absmax),
                        deq=sha(deq), packed_numel=packed.numel(), absmax_numel=st.absmax.numel())
    # rowwise int8 on the same matrix (config 4 operand)
    q, s = ref.quantize_rowwise(x)
    out["A_rowwise"] = dict(shape=[4096, 4096], dtype="f16", seed=1234, std=1.0, q=sha(q), scales=sha(s),
                            deq=sha(ref.dequantize_rowwise(q, s, torch.float16)))
    del x, packed, deq, q, s
    # config B: 11008x4096 bf16 ~N(0,0.02^2), NF4 + double quant (BASELINE configs[2])
    x = synthetic.normal((11008, 4096), torch.bfloat16, seed=1235, std=0.02)
    packed, st = ref.quantize_nf4(x, blocksize=64, compress_statistics=True)
    deq = ref.dequantize_nf4(packed, st)
    out["B"] = dict(shape=[11008, 4096], dtype="bf16", seed=1235, std=0.02, blocksize=64, quant_type="nf4",
                    compress_statistics=True, input=sha(x), packed=sha(packed), absmax=sha(st.absmax),
                    absmax2=sha(st.state2.absmax), deq=sha(deq), packed_numel=packed.numel(),
                    absmax_numel=st.absmax.numel(), absmax2_numel=st.state2.absmax.numel())
    out["seconds"] = round(time.time() - t0, 1)
    with open(os.path.join(HERE, "g3_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    manifest["g3"] = sorted(k for k in out if k != "seconds")


# ----------------------------------------------------------------------------- G4
def make_g4(manifest):
    arrays = {}
    cases = []
    mnk = [(1, 64, 128), (7, 64, 128), (32, 63, 127), (32, 64, 65), (128, 256, 512), (3, 17, 70), (16, 128, 256)]
    i = 0
    for (M, N, K) in mnk:
        for qt in ("nf4", "fp4"):
            wdt = ["f16", "bf16", "f32"][i % 3]
            adt = ["f16", "bf16", "f16", "f32"][i % 4]
            bdt = [None, "f16", "f32", "bf16"][i % 4]
            cs = bool(i % 2)
            bs = [64, 32, 128][i % 3]
            cd = [None, "f16", "bf16"][(i // 2) % 3]
            cases.append((M, N, K, qt, wdt, adt, bdt, cs, bs, cd))
            i += 1
    # the batched [B,S,K] form of Linear4bit.forward (nn/linear4bit.py:106-117)
    cases.append(((2, 5), 64, 128, "nf4", "f16", "f16", "f16", False, 64, None))
    for ci, (M, N, K, qt, wdt, adt, bdt, cs, bs, cd) in enumerate(cases):
        seed = 400 + 3 * ci
        W = synthetic.normal((N, K), DT[wdt], seed=seed)
        lead = M if isinstance(M, tuple) else (M,)
        A = synthetic.normal(lead + (K,), DT[adt], seed=seed + 1)
        bias = None if bdt is None else synthetic.normal((N,), DT[bdt], seed=seed + 2)
        packed, st = ref.quantize_4bit(W, blocksize=bs, compress_statistics=cs, quant_type=qt)
        out = ref.matmul_4bit(A, packed, st, bias, None if cd is None else DT[cd])
        key = f"c{ci}_"
        arrays[key + "W"] = bits(W)
        arrays[key + "A"] = bits(A)
        if bias is not None:
            arrays[key + "bias"] = bits(bias)
        arrays[key + "packed"] = bits(packed)
        state_arrays(key, st, arrays)
        arrays[key + "out"] = bits(out)
        manifest["g4"].append(dict(id=ci, M=list(lead), N=N, K=K, quant_type=qt, w_dtype=wdt, a_dtype=adt,
                                   bias_dtype=bdt, compress_statistics=cs, blocksize=bs, compute_dtype=cd,
                                   out_dtype={torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[out.dtype],
                                   seed=seed))
    np.savez_compressed(os.path.join(HERE, "g4_matmul.npz"), **arrays)


# ----------------------------------------------------------------------------- G5
def make_g5(manifest):
    arrays = {}
    g5 = manifest["g5"]
    # quantize_rowwise / dequantize_rowwise
    for ci, (shape, dt) in enumerate([((64, 128), "f16"), ((7, 13), "bf16"), ((2, 3, 32), "f32"), ((1, 1), "f16")]):
        x = synthetic.normal(shape, DT[dt], seed=500 + ci)
        q, s = ref.quantize_rowwise(x)
        arrays[f"rw{ci}_x"] = bits(x)
        arrays[f"rw{ci}_q"] = bits(q)
        arrays[f"rw{ci}_s"] = bits(s)
        for odt in ("f16", "bf16", "f32"):
            arrays[f"rw{ci}_deq_{odt}"] = bits(ref.dequantize_rowwise(q, s, DT[odt]))
        g5.append(dict(kind="rowwise", id=ci, shape=list(shape), dtype=dt))
    # the known-answer case of tests/test_advanced_linear.py:139-153: fill(0.5) -> 127
    x = torch.full((8, 32), 0.5, dtype=torch.float16)
    q, s = ref.quantize_rowwise(x)
    arrays["rwfill_x"], arrays["rwfill_q"], arrays["rwfill_s"] = bits(x), bits(q), bits(s)
    # rounding ties: x * (127/absmax) lands on .5 exactly for absmax = 127
    x = torch.tensor([[127.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, -126.5, 3.5, 0.25, 0.75]], dtype=torch.float32)
    q, s = ref.quantize_rowwise(x)
    arrays["rwtie_x"], arrays["rwtie_q"], arrays["rwtie_s"] = bits(x), bits(q), bits(s)
    z = torch.zeros(3, 16, dtype=torch.float16)
    q, s = ref.quantize_rowwise(z)
    arrays["rwzero_x"], arrays["rwzero_q"], arrays["rwzero_s"] = bits(z), bits(q), bits(s)
    # quantize_blockwise / dequantize_blockwise (a4/a5), incl. ragged tail and nested
    for ci, (n, dt, bs, nested) in enumerate([(1000, "f32", 256, False), (4096, "f16", 4096, False),
                                               (70000, "f32", 256, True), (513, "bf16", 64, False)]):
        x = synthetic.normal((n,), DT[dt], seed=520 + ci).abs() if ci == 0 else synthetic.normal((n,), DT[dt], seed=520 + ci)
        q, st = ref.quantize_blockwise(x, blocksize=bs, nested=nested)
        arrays[f"bw{ci}_x"] = bits(x)
        arrays[f"bw{ci}_q"] = bits(q)
        arrays[f"bw{ci}_absmax"] = bits(st.absmax)
        if nested:
            arrays[f"bw{ci}_absmax2"] = bits(st.state2.absmax)
        arrays[f"bw{ci}_deq"] = bits(ref.dequantize_blockwise(q, st))
        g5.append(dict(kind="blockwise", id=ci, numel=n, dtype=dt, blocksize=bs, nested=nested))
    # matmul_int8
    for ci, (M, K, N, odt) in enumerate([(32, 64, 48, "f16"), (5, 33, 7, "f16"), (64, 128, 64, "bf16"), (16, 256, 32, "f32")]):
        A = synthetic.normal((M, K), torch.float16, seed=540 + 2 * ci)
        B = synthetic.normal((K, N), torch.float16, seed=541 + 2 * ci)
        Aq, As = ref.quantize_rowwise(A)
        Bq_t, Bs = ref.quantize_rowwise(B.t().contiguous())
        Bq = Bq_t.t().contiguous()
        out = ref.matmul_int8(Aq, Bq, As, Bs, DT[odt])
        arrays[f"mm{ci}_A"], arrays[f"mm{ci}_B"] = bits(Aq), bits(Bq)
        arrays[f"mm{ci}_As"], arrays[f"mm{ci}_Bs"] = bits(As), bits(Bs)
        arrays[f"mm{ci}_out"] = bits(out)
        g5.append(dict(kind="matmul_int8", id=ci, M=M, K=K, N=N, dtype=odt))
    # double_quant (a13)
    for ci, (shape, dt) in enumerate([((8, 16), "f16"), ((33, 70), "f32"), ((64, 64), "bf16")]):
        x = synthetic.normal(shape, DT[dt], seed=560 + ci)
        oc, orow, cs, rs, _ = ref.double_quant(x)
        arrays[f"dq{ci}_x"] = bits(x)
        arrays[f"dq{ci}_out_col"], arrays[f"dq{ci}_out_row"] = bits(oc), bits(orow)
        arrays[f"dq{ci}_col_stats"], arrays[f"dq{ci}_row_stats"] = bits(cs), bits(rs)
        g5.append(dict(kind="double_quant", id=ci, shape=list(shape), dtype=dt))
    # Linear8bit.forward (a15)
    for ci, (M, K, N, dt, has_bias) in enumerate([(4, 64, 32, "f16", True), ((2, 3), 128, 48, "bf16", False), (9, 70, 33, "f16", True)]):
        lin = torch.nn.Linear(K, N, bias=has_bias)
        with torch.no_grad():
            lin.weight.copy_(synthetic.normal((N, K), torch.float32, seed=580 + 3 * ci, std=0.05))
            if has_bias:
                lin.bias.copy_(synthetic.normal((N,), torch.float32, seed=581 + 3 * ci))
        lin = lin.to(DT[dt])
        l8 = ref.Linear8bit.from_linear(lin)
        lead = M if isinstance(M, tuple) else (M,)
        x = synthetic.normal(lead + (K,), DT[dt], seed=582 + 3 * ci)
        y = l8(x)
        arrays[f"l8{ci}_W"] = bits(lin.weight.data)
        if has_bias:
            arrays[f"l8{ci}_bias"] = bits(lin.bias.data)
        arrays[f"l8{ci}_x"], arrays[f"l8{ci}_y"] = bits(x), bits(y)
        arrays[f"l8{ci}_q"], arrays[f"l8{ci}_s"] = bits(l8.weight_int8), bits(l8.weight_scales)
        g5.append(dict(kind="linear8bit", id=ci, M=list(lead), K=K, N=N, dtype=dt, bias=has_bias))
    np.savez_compressed(os.path.join(HERE, "g5_int8.npz"), **arrays)


# ----------------------------------------------------------------------------- API shape capture
def make_api(manifest):
    """Names / keys / messages the Python counterpart must reproduce."""
    lin = torch.nn.Linear(70, 5).half()
    l4 = ref.Linear4bit.from_linear(lin, compress_statistics=True)
    sd = l4.state_dict()
    api = dict(
        linear4bit_state_dict_keys=sorted(sd.keys()),
        quant_state_dict_keys=sorted(sd["weight_quant_state"].keys()),
        state2_dict_keys=sorted(sd["weight_quant_state"]["state2"].keys()),
        linear4bit_weight_numel=int(sd["weight"].numel()),
        linear4bit_ctor_weight_numel=int(ref.Linear4bit(70, 5).weight.numel()),
        linear8bit_state_dict_keys=sorted(ref.Linear8bit.from_linear(lin).state_dict().keys()),
    )
    msgs = {}
    x = torch.zeros(4, 64)
    for name, kw in (("neg", dict(blocksize=-1)), ("zero", dict(blocksize=0)), ("large", dict(blocksize=131072)),
                     ("pow2", dict(blocksize=48)), ("qtype", dict(quant_type="int4"))):
        try:
            ref.quantize_4bit(x, **kw)
        except ValueError as e:
            msgs[name] = str(e)
    try:
        ref.quantize_blockwise(x, blocksize=0)
    except ValueError as e:
        msgs["blockwise_zero"] = str(e)
    api["error_messages"] = msgs
    manifest["api"] = api


def main():
    torch.manual_seed(0)
    manifest = dict(
        provenance=dict(reference="mpsops/mps-bitsandbytes v%s (/root/reference, CPU fallback path)" % ref.__version__,
                        torch=torch.__version__, numpy=np.__version__,
                        generated=time.strftime("%Y-%m-%d"), script="tests/golden/make_golden.py"),
        g1=[], g2=[], g3=[], g4=[], g5=[])
    make_g1(manifest)
    make_g2(manifest)
    make_g4(manifest)
    make_g5(manifest)
    make_api(manifest)
    make_g3(manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
