"""Host-side logic of the Python mirror (no GPU): QuantState wire format, validation messages
(the reference's tests match them by regex, tests/test_fp4_fp8_double.py:414-460), shape
arithmetic, module construction and state-dict keys — pinned to what the reference produced
(tests/golden/manifest.json 'api')."""
import pytest
import torch

import mps_bitsandbytes_amd as bnb
from mps_bitsandbytes_amd import functional as F
from mps_bitsandbytes_amd.functional import QuantState
from mps_bitsandbytes_amd.sharding import row_shard


def test_codebooks_match_reference_tables():
    assert F.NF4_CODEBOOK.dtype == torch.float32 and F.NF4_CODEBOOK.numel() == 16
    assert F.NF4_CODEBOOK[7].item() == 0.0 and F.NF4_CODEBOOK[0].item() == -1.0 and F.NF4_CODEBOOK[15].item() == 1.0
    assert torch.equal(torch.sort(F.NF4_CODEBOOK).values, F.NF4_CODEBOOK)
    assert F.FP4_CODEBOOK.tolist()[:8] == [0.0, 0.0625, 0.125, 0.25, 0.375, 0.5, 0.75, 1.0]
    assert torch.equal(F.FP4_CODEBOOK[8:], -F.FP4_CODEBOOK[:8])
    assert torch.equal(bnb.create_normal_map(), F.NF4_CODEBOOK) and torch.equal(bnb.create_fp4_map(), F.FP4_CODEBOOK)


def test_padding_rule():
    assert F._padded(4096, 64) == 4096
    assert F._padded(70, 64) == 128
    assert F._padded(13, 64) == 64
    assert F._padded(3, 1) == 4       # odd -> + blocksize (functional.py:220-221)
    assert F._padded(1, 1) == 2


def test_quantstate_dict_roundtrip(golden):
    st2 = QuantState(absmax=torch.ones(3), shape=torch.Size([700]), blocksize=256, quant_type="int8", dtype=torch.float32)
    st = QuantState(absmax=torch.zeros(700, dtype=torch.int8), shape=torch.Size([10, 70]), blocksize=64,
                    quant_type="fp4", dtype=torch.bfloat16, state2=st2)
    assert torch.equal(st.code, F.FP4_CODEBOOK)
    d = st.as_dict()
    assert sorted(d.keys()) == golden.manifest["api"]["quant_state_dict_keys"]
    assert sorted(d["state2"].keys()) == golden.manifest["api"]["state2_dict_keys"]
    back = QuantState.from_dict(d)
    assert back.blocksize == 64 and back.quant_type == "fp4" and back.dtype == torch.bfloat16
    assert back.state2.blocksize == 256 and torch.equal(back.state2.absmax, st2.absmax)
    assert QuantState(absmax=torch.ones(1), shape=torch.Size([1])).quant_type == "nf4"


def test_validation_messages_match_reference(golden):
    msgs = golden.manifest["api"]["error_messages"]
    x = torch.zeros(4, 64)
    for name, kw in (("neg", dict(blocksize=-1)), ("zero", dict(blocksize=0)), ("large", dict(blocksize=131072)),
                     ("pow2", dict(blocksize=48)), ("qtype", dict(quant_type="int4"))):
        with pytest.raises(ValueError) as e:
            # quant_type is validated before the device, blocksize after it (reference order):
            # use a meta 'cuda' check bypass by validating on the message only for quant_type
            if name == "qtype":
                bnb.quantize_4bit(x, **kw)
            else:
                _validate_blocksize_like_reference(**kw)
        assert str(e.value) == msgs[name]


def _validate_blocksize_like_reference(blocksize):
    """quantize_4bit's blocksize checks sit behind the device gate; exercise them through a
    tensor whose device type reads 'cuda' without needing a GPU."""
    class FakeCuda(torch.Tensor):
        @property
        def device(self):
            return torch.device("cuda", 0)
    t = torch.zeros(4, 64).as_subclass(FakeCuda)
    return bnb.quantize_4bit(t, blocksize=blocksize)


def test_blockwise_validation_message(golden):
    class FakeCuda(torch.Tensor):
        @property
        def device(self):
            return torch.device("cuda", 0)
    with pytest.raises(ValueError) as e:
        bnb.quantize_blockwise(torch.zeros(8).as_subclass(FakeCuda), blocksize=0)
    assert str(e.value) == golden.manifest["api"]["error_messages"]["blockwise_zero"]


def test_linear4bit_construction_and_state_dict_keys(golden):
    api = golden.manifest["api"]
    l4 = bnb.Linear4bit(70, 5)
    assert l4.weight.dtype == torch.uint8 and l4.weight.numel() == api["linear4bit_ctor_weight_numel"]
    assert l4.bias.dtype == torch.float16 and l4.weight_quant_state is None and l4.quant_state is None
    with pytest.raises(RuntimeError, match="Weight not quantized"):
        l4(torch.zeros(1, 70))
    with pytest.raises(ValueError, match="quant_type must be"):
        bnb.Linear4bit(8, 8, quant_type="int4")
    # a checkpoint produced elsewhere loads on CPU without running any kernel
    st2 = QuantState(absmax=torch.ones(1), shape=torch.Size([10]), blocksize=256, quant_type="int8", dtype=torch.float32)
    st = QuantState(absmax=torch.zeros(10, dtype=torch.int8), shape=torch.Size([5, 70]), blocksize=64, state2=st2)
    l4.weight = torch.zeros(api["linear4bit_weight_numel"], dtype=torch.uint8)
    l4.weight_quant_state = st
    sd = l4.state_dict()
    assert sorted(sd.keys()) == api["linear4bit_state_dict_keys"]
    fresh = bnb.Linear4bit(70, 5, blocksize=128, quant_type="fp4")
    with pytest.warns(UserWarning, match="mismatch"):
        fresh.load_state_dict(sd)
    assert fresh.blocksize == 64 and fresh.quant_type == "nf4"
    assert fresh.weight.numel() == api["linear4bit_weight_numel"]
    assert fresh.weight_quant_state.state2.blocksize == 256
    assert "quant_type=nf4" in repr(fresh)
    assert fresh.device.type == "cpu"


def test_linear8bit_construction(golden):
    l8 = bnb.Linear8bit(32, 16, bias=True, compute_dtype=torch.bfloat16)
    assert sorted(l8.state_dict().keys()) == golden.manifest["api"]["linear8bit_state_dict_keys"]
    assert l8.weight_int8.shape == (16, 32) and l8.weight_int8.dtype == torch.int8
    assert l8.weight_scales.dtype == torch.float32 and l8.bias.dtype == torch.bfloat16
    l8.clear_cache()
    assert l8._weight_cache is None and l8.device.type == "cpu"


def test_row_shard_partitions_exactly():
    for M, w in ((32768, 8), (4096, 1), (10, 4), (3, 8), (0, 2)):
        spans = [row_shard(M, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == M
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_shard(8, 2, 2)


def test_bitsandbytes_config_mirrors_reference():
    cfg = bnb.BitsAndBytesConfig(load_in_4bit=True, bnb_4bit_quant_type="fp4", bnb_4bit_compute_dtype=torch.bfloat16,
                                 bnb_4bit_use_double_quant=True)
    assert cfg.is_quantizable and cfg.quantization_method == "bitsandbytes_4bit" and cfg.llm_int8_skip_modules == []
    d = cfg.to_dict()
    assert sorted(d) == ["bnb_4bit_compute_dtype", "bnb_4bit_quant_type", "bnb_4bit_use_double_quant", "llm_int8_skip_modules",
                         "llm_int8_threshold", "load_in_4bit", "load_in_8bit"]
    back = bnb.BitsAndBytesConfig.from_dict(d)
    # the reference's string parse tests 'float16' first and 'torch.bfloat16' contains it (reference integration.py:86-92):
    # the string round trip lands on float16 there, and therefore here; a torch.dtype value passes through
    assert back.bnb_4bit_compute_dtype == torch.float16 and back.bnb_4bit_quant_type == "fp4" and back.load_in_4bit
    d2 = dict(d, bnb_4bit_compute_dtype=torch.bfloat16)
    assert bnb.BitsAndBytesConfig.from_dict(d2).bnb_4bit_compute_dtype == torch.bfloat16
    assert bnb.BitsAndBytesConfig.from_dict(dict(d, bnb_4bit_compute_dtype="float32")).bnb_4bit_compute_dtype == torch.float16
    assert bnb.BitsAndBytesConfig().quantization_method == "none"
    # the coercion of a bf16 / f32 NAME to fp16 is kept (reference behaviour) but is never silent
    import warnings
    from mps_bitsandbytes_amd import integration
    integration._dtype_coercion_warned = False
    with pytest.warns(RuntimeWarning, match="parses to torch.float16"):
        bnb.BitsAndBytesConfig.from_dict(dict(d, bnb_4bit_compute_dtype="torch.bfloat16"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")     # a float16 name is what it says: no warning
        assert bnb.BitsAndBytesConfig.from_dict(dict(d, bnb_4bit_compute_dtype="torch.float16")).bnb_4bit_compute_dtype == torch.float16
    with pytest.raises(ValueError, match="both 4-bit and 8-bit"):
        bnb.BitsAndBytesConfig(load_in_4bit=True, load_in_8bit=True)
    with pytest.raises(ValueError, match="bnb_4bit_quant_type must be"):
        bnb.BitsAndBytesConfig(bnb_4bit_quant_type="int4")


def test_memory_footprint_counts_quantized_buffers():
    m = torch.nn.Sequential(bnb.Linear4bit(64, 32, bias=False), bnb.Linear8bit(32, 16, bias=True))
    fp = bnb.get_memory_footprint(m)
    assert fp["quantized_params"] == 64 * 32 // 2 + 32 * 16
    assert fp["total_params"] > 0 and fp["actual_size_gb"] > 0


def test_embedding_and_outlier_modules_mirror_reference_without_a_gpu():
    """Constructors, buffers, validation messages and the no-CPU-path rule of the §8f rank-3 modules."""
    e4 = bnb.Embedding4bit(100, 64, padding_idx=3, quant_type="fp4", blocksize=32)
    assert sorted(e4.state_dict()) == ["weight_absmax", "weight_packed"]
    assert e4.weight_packed.shape == (100, 32) and e4.weight_packed.dtype == torch.uint8 and e4.weight_absmax.shape == (100, 2)
    assert "quant_type=fp4" in e4.extra_repr()
    with pytest.raises(ValueError, match="embedding_dim must be even"):
        bnb.Embedding4bit(10, 15)
    with pytest.raises(ValueError, match="quant_type must be 'nf4' or 'fp4'"):
        bnb.Embedding4bit(10, 16, quant_type="int4")
    e8 = bnb.Embedding8bit(100, 70, padding_idx=0)
    assert sorted(e8.state_dict()) == ["weight_int8", "weight_scales"] and e8.weight_int8.shape == (100, 70)
    assert bnb.EmbeddingNF4(10, 16).quant_type == "nf4" and bnb.EmbeddingFP4(10, 16).quant_type == "fp4"
    oa = bnb.OutlierAwareLinear(64, 32, bias=True, threshold=5.0)
    assert sorted(oa.state_dict()) == ["bias", "outlier_indices", "outlier_weights", "weight_int8", "weight_scales"]
    assert oa.outlier_weights.shape == (32, 0) and oa.outlier_indices.dtype == torch.long and "outliers=0" in oa.extra_repr()
    assert bnb.OutlierAwareLinear(8, 4, bias=False).bias is None
    # no CPU path: CPU tensors are rejected, nothing falls back
    for fn in (lambda: e4(torch.tensor([1, 2])), lambda: e8(torch.tensor([1, 2])), lambda: oa(torch.zeros(2, 64)),
               lambda: bnb.Embedding4bit.from_embedding(torch.nn.Embedding(8, 64).half()),
               lambda: bnb.OutlierAwareLinear.from_linear(torch.nn.Linear(64, 8).half())):
        with pytest.raises(ValueError, match="requires tensor on a 'cuda'"):
            fn()


def test_linear_fp8_module_mirrors_reference_without_a_gpu():
    l = bnb.LinearFP8(64, 32, bias=True, compute_dtype=torch.bfloat16)
    assert sorted(l.state_dict()) == ["bias", "weight_fp8", "weight_scales"]
    assert l.weight_fp8.dtype == torch.uint8 and l.weight_fp8.shape == (32, 64) and l.bias.dtype == torch.bfloat16
    assert "quant_type=fp8_e4m3" in l.extra_repr() and bnb.LinearFP8(8, 4, bias=False).bias is None
    with pytest.raises(ValueError, match="Input must be 2D"):
        bnb.quantize_fp8_e4m3(torch.zeros(4))
    for fn in (lambda: bnb.quantize_fp8_e4m3(torch.zeros(4, 8)), lambda: l(torch.zeros(2, 64)),
               lambda: bnb.LinearFP8.from_linear(torch.nn.Linear(64, 8).half())):
        with pytest.raises(ValueError, match="requires tensor on a 'cuda'"):
            fn()


def test_absmax_descriptor_never_hands_the_kernels_a_host_pointer_or_a_short_absmax():
    """the C ABI receives no lengths and dereferences what it is given; the host mirror must stop a CPU-resident
    absmax and an absmax whose count disagrees with shape/blocksize BEFORE any launch (the reference raises from
    absmax.view(N, num_blocks_per_row), functional.py:371-373)."""
    from mps_bitsandbytes_amd.functional import _absmax_desc, _check_absmax_count
    with pytest.raises(ValueError, match="cuda"):
        _absmax_desc(torch.zeros(4), None, [])
    with pytest.raises(ValueError, match="absmax has 7 elements, expected 8"):
        _check_absmax_count(torch.zeros(7), 2, 256, 64, "matmul_4bit")
    _check_absmax_count(torch.zeros(8), 2, 256, 64, "matmul_4bit")


def test_synthetic_device_form_equals_the_numpy_form():
    """bench.py draws its inputs with synthetic.normal_device (torch integer ops, runs on the GPU); it must give the bits of
    synthetic.normal (numpy), which the goldens and the parity tests use."""
    from mps_bitsandbytes_amd import synthetic
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for seed, std, shape in ((1234, 1.0, (257, 1031)), (4321, 0.02, (5, 70)), (7, 3.5, (100000,))):
            assert torch.equal(synthetic.normal(shape, dt, seed=seed, std=std), synthetic.normal_device(shape, dt, seed=seed, std=std, device="cpu"))


def test_stream_and_device_helpers_accept_what_torch_accepts(monkeypatch):
    """_native.stream_ptr / on_device take a torch.device, a device string or an index (round 4: the raw-stream fast path read `.index` off whatever it
    was given -- on the string "cuda" that is str.index)."""
    from mps_bitsandbytes_amd import _native
    seen = []
    monkeypatch.setattr(_native, "_raw_stream", lambda i: seen.append(i) or 1234)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    for d in ("cuda", "cuda:0", torch.device("cuda"), torch.device("cuda", 0), 0):
        assert _native.stream_ptr(d).value == 1234
        assert _native.on_device(d) is _native._NO_GUARD
    assert seen == [0, 0, 0, 0, 0]
    assert _native.on_device(torch.device("cuda", 1)) is not _native._NO_GUARD      # another device: a real guard
