"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mbnb_hip.h declares (no compute calls — there is no GPU here), and argument errors are
reported through the status / last-error convention without touching a device."""
import ctypes
import os
import re

import pytest
import torch

from mps_bitsandbytes_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mbnb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbnb_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _native.lib()
    names = _declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mbnb_hip.h but not exported"
    assert sorted(_native.EXPORTED_SYMBOLS) == names, "python binding and header disagree"
    assert len(names) <= 30, "ABI version 2: one entry point per operation (+ its workspace query)"
    assert lib.mbnb_abi_version() == _native.ABI_VERSION == 2
    header = open(os.path.join(ROOT, "include", "mbnb_hip.h")).read()
    assert re.search(r"#define MBNB_ABI_VERSION 2\b", header)


def test_library_exports_nothing_undeclared():
    """Every `mbnb_*` C symbol of the shared object is declared in the header (no leftover version-1 entry points)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(mbnb_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == _declared_symbols()


def test_argument_errors_use_status_and_last_error():
    lib = _native.lib()
    rc = lib.mbnb_quantize_4bit(None, 0, 4, 64, 64, 48, 0, None, None, None, None)
    assert rc == -1 and b"power of 2" in lib.mbnb_last_error()
    rc = lib.mbnb_quantize_4bit(None, 0, 4, 64, 100, 64, 0, None, None, None, None)
    assert rc == -2 and b"cols_padded" in lib.mbnb_last_error()
    rc = lib.mbnb_quantize_4bit(None, 7, 4, 64, 64, 64, 0, None, None, None, None)
    assert rc == -1
    rc = lib.mbnb_matmul_4bit(None, 4, 64, None, None, 8, 64, 64, 0, 0, None, 0, None, None, 0, 0, None)
    assert rc == -1 and b"absmax" in lib.mbnb_last_error()
    desc = _native.AbsmaxDesc(None, 16, None, 256)  # int8 absmax without absmax2
    rc = lib.mbnb_dequantize_4bit(None, ctypes.byref(desc), 1, 64, 64, 64, 0, 0, None, None)
    assert rc == -1 and b"absmax2" in lib.mbnb_last_error()
    # empty problems are a no-op success
    assert lib.mbnb_quantize_rowwise(None, 0, 0, 128, None, None, None) == 0
    assert lib.mbnb_matmul_int8(None, None, None, None, 0, 16, 16, 0, None, None, 0, None) == 0
    with pytest.raises(RuntimeError, match="status -1"):
        _native.check(-1, "unit")
    # the nn entry points validate before touching the device too
    assert lib.mbnb_embedding_4bit(None, 4, None, None, 10, 15, 64, 0, 0, 0, 0, None, None) == -1 and b"even" in lib.mbnb_last_error()
    assert lib.mbnb_embedding_4bit(None, 0, None, None, 10, 16, 64, 0, 0, 0, 0, None, None) == 0
    assert lib.mbnb_embedding_8bit(None, 4, None, None, 10, 16, 0, 0, 0, None, None) == -1
    assert lib.mbnb_outlier_linear(None, 0, 4, 64, None, None, 8, None, 2, None, None, None, None, 0, None) == -1


def test_gemm_dense_and_matmul_ex_argument_errors():
    """The round-2 entry points validate before touching the device."""
    lib = _native.lib()
    one = ctypes.c_void_p(256)    # any non-NULL, 16-byte aligned value: validation must fail before a dereference
    assert lib.mbnb_gemm_dense(one, one, 2, None, 1, one, 256, 256, 256, 256, None, 0, 1, None) == -1          # f32 weights
    assert b"dtype" in lib.mbnb_last_error()
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 200, 200, None, 0, 1, None) == -2          # K % 64
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 256, 128, None, 0, 1, None) == -2          # ldw < K
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 256, 256, None, 0, 2, None) == -1          # split without workspace
    assert b"workspace" in lib.mbnb_last_error()
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 256, 256, None, 0, 1 | (4 << 8), None) == -1   # tile code 4 (3 = 128 x 128 since round 3)
    assert b"tile code" in lib.mbnb_last_error()
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 256, 256, one, 1 << 30, 2 | (3 << 8), None) == -1   # the 128 x 128 tile takes no K slices
    assert lib.mbnb_gemm_dense(one, one, 1, None, 1, one, 256, 256, 128, 128, None, 0, 1 | (3 << 8), None) == -1        # ... and needs K >= 192
    assert lib.mbnb_gemm_dense(ctypes.c_void_p(8), one, 1, None, 1, one, 256, 256, 256, 256, None, 0, 1, None) == -1  # misaligned A
    rc = lib.mbnb_matmul_4bit(one, 4, 64, one, None, 8, 64, 64, 0, 0, None, 0, one, None, 0, 0, None)
    assert rc == -1 and b"absmax" in lib.mbnb_last_error()
    for bad in (2, 6, 32):    # version 1's FUSED4 / sync-only flags are gone: MBNB_MATMUL_FUSED_ONLY is the only bit
        rc = lib.mbnb_matmul_4bit(one, 4, 64, one, None, 8, 64, 64, 0, 0, None, 0, one, None, 0, bad, None)
        assert rc == -1 and b"flags" in lib.mbnb_last_error()
        assert lib.mbnb_linear_int8(one, 0, 4, 64, one, one, 8, None, one, None, 0, bad, None) == -1 and b"flags" in lib.mbnb_last_error()
        assert lib.mbnb_linear_fp8(one, 0, 4, 64, one, one, 8, None, one, None, 0, bad, None) == -1 and b"flags" in lib.mbnb_last_error()
    assert lib.mbnb_matmul_4bit(one, 4, 64, one, None, 8, 64, 64, 0, 0, None, 0, one, None, -1, 0, None) == -1   # negative workspace size


def test_workspace_size_functions_are_pure_host_code():
    """Workspace policy.  Split-K share: nothing for GEMV / skinny-sized M or for shapes that fill the chip with 256 x 256
    tiles; slices x tiles x 64 KiB (128 x 128 f32) in between.  Full query: from 256 rows and 1.5 M outputs up the
    dequantised weight (N x K_weight x 2 bytes, 256-byte granules) plus slices x M x N f32 partials."""
    lib = _native.lib()
    FUSED_ONLY = 1

    def sk(M, N, K):      # the split-K share alone = the query of a caller that keeps the fused kernels
        return lib.mbnb_matmul_4bit_workspace_bytes(M, N, K, K, 1, FUSED_ONLY)

    def full(M, N, K):
        return lib.mbnb_matmul_4bit_workspace_bytes(M, N, K, K, 1, 0)

    assert sk(1, 4096, 4096) == 0
    assert sk(4, 4096, 4096) == 0
    assert sk(4096, 4096, 4096) == 0                        # 256 tiles of 256^2
    assert sk(128, 4096, 4096) == 16 * 32 * 65536           # 32 tiles -> 16 slices of 256 k
    assert sk(1024, 4096, 4096) == 2 * 256 * 65536
    assert sk(128, 4096, 72) == 0                           # K % 64 != 0
    assert sk(0, 4096, 4096) == 0
    assert lib.mbnb_linear_int8_workspace_bytes(200, 4096, 4096, 0) == sk(200, 4096, 4096) > 0
    assert lib.mbnb_linear_int8_workspace_bytes(4096, 4096, 4096, 0) == 4096 * 4096 * 2
    assert lib.mbnb_linear_int8_workspace_bytes(4096, 4096, 4096, FUSED_ONLY) == 0      # no scratch for the weight: fused W8A16 kernels
    assert full(128, 4096, 4096) == sk(128, 4096, 4096)     # below 512 rows: the split-K share only
    assert full(256, 4096, 4096) == sk(256, 4096, 4096)     # 1.05 M outputs: not yet
    assert full(384, 4096, 4096) > 4096 * 4096 * 2          # 1.57 M outputs: the weight + split-K partials
    assert full(4096, 4096, 4096) == 4096 * 4096 * 2        # the dequantised weight, no split (256 tiles)
    assert full(32768, 4096, 4096) == 4096 * 4096 * 2
    assert full(1024, 4096, 4096) == 4096 * 4096 * 2        # round 3: 256 tiles of 128 x 128 in one round (gemm_dense128.h) instead of 2 K slices
    assert full(512, 2048, 8192) == 2048 * 8192 * 2 + 8 * 512 * 2048 * 4   # 64 tiles of 128 x 128, long K: 8 slices of f32 partials
    assert full(2048, 4096, 4096) == 4096 * 4096 * 2        # from 96 tiles up: never split (row bits independent of M)
    assert full(4096, 4096, 4104) == sk(4096, 4096, 4104)   # K % 64 != 0: fused kernels only
    q = lib.mbnb_matmul_4bit_workspace_bytes
    assert q(4096, 4096, 4096, 4096, 0, 0) == q(4096, 4096, 4096, 4096, 1, 0) == full(4096, 4096, 4096)   # f16 and bf16 alike
    assert q(4096, 1000, 192, 256, 1, 0) == ((1000 * 256 * 2 + 255) // 256) * 256   # padded weight rows
    assert q(4096, 4096, 4096, 4095, 1, 0) == 0             # K_weight < K: not a weight
    assert q(4096, 4096, 4096, 4096, 7, 0) == 0             # not a dtype
    assert q(4096, 4096, 4096, 4096, 2, 0) == 4096 * 4096 * 4   # f32 weight dtype: the weight dequantised once as f32
    assert q(4096, 4096, 4096, 4096, 2, FUSED_ONLY) == 0
    assert q(17, 4096, 4096, 4096, 2, 0) == 4096 * 4096 * 4 + 16 * 17 * 4096 * 4   # 64 tiles of 64 x 64: 16 K slices of f32 partials
    assert q(16, 4096, 4096, 4096, 2, 0) == 0                   # two 8-row chunks of the generic kernel are cheaper
    assert q(64, 1024, 1024, 1024, 2, 0) == 0 and q(128, 1024, 1024, 1024, 2, 0) > 0   # small layers: later
    assert q(64, 256, 130, 192, 2, 0) == 0                      # K % 4 != 0
    assert q(300, 1000, 1028, 1088, 2, 0) == 1000 * 1088 * 4 + 4 * 300 * 1000 * 4
    assert lib.mbnb_outlier_linear_workspace_bytes(4096, 4096, 0) == lib.mbnb_outlier_linear_workspace_bytes(4096, 4096, 16) == 4096 * 4096 + 4 * 4096 + 32 * 4096
    assert lib.mbnb_outlier_linear_workspace_bytes(3, 5, 2) == 256 + 256 + 256
    assert lib.mbnb_outlier_linear_workspace_bytes(4096, 4096, 40) > lib.mbnb_outlier_linear_workspace_bytes(4096, 4096, 16)
    assert lib.mbnb_matmul_int8_workspace_bytes(4096, 4096, 4096) == 0        # B read in place
    assert lib.mbnb_matmul_int8_workspace_bytes(100, 200, 304) == 200 * 304


def test_product_has_no_cpu_path_and_never_imports_the_oracle():
    import mps_bitsandbytes_amd as bnb
    with pytest.raises(ValueError, match="requires tensor on a 'cuda'"):
        bnb.quantize_nf4(torch.zeros(4, 64))
    with pytest.raises(ValueError, match="requires tensor on a 'cuda'"):
        bnb.quantize_rowwise(torch.zeros(4, 64))
    with pytest.raises(ValueError, match="requires tensor on a 'cuda'"):
        bnb.dequantize_blockwise(torch.zeros(4, dtype=torch.int8), absmax=torch.ones(1))
    pkg = os.path.join(ROOT, "mps_bitsandbytes_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src, f"{f} references the oracle library"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_load_error", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no Python fallback"):
        _native.lib()
    assert _native.available() is False
