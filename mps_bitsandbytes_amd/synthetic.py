"""
Deterministic synthetic inputs for tests, goldens and the bench driver.

Values come from a counter-based integer hash (splitmix64 finaliser) and an
Irwin-Hall(12) sum of 16-bit uniforms, i.e. *integer arithmetic only* up to one
final float64 multiply — no libm, no torch RNG — so the build container (where
the golden vectors are captured by running the reference) and the GPU box
regenerate bit-identical inputs without shipping the tensors
(SURVEY.md §7 step 1, §8d "Synthetic inputs").
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def normal_f64(numel: int, seed: int, offset: int = 0) -> np.ndarray:
    """~N(0,1) samples (Irwin-Hall 12, support ±6), float64, element i depends only on (seed, offset+i)."""
    out = np.empty(numel, dtype=np.float64)
    chunk = 1 << 22
    # std of a uniform integer on [0, 65535] is sqrt((65536^2 - 1) / 12); 12 of them sum to variance 65536^2 - 1
    inv_std = 1.0 / np.sqrt(65536.0 * 65536.0 - 1.0)
    mean = 12 * 32767.5
    with np.errstate(over="ignore"):
        for s in range(0, numel, chunk):
            e = min(numel, s + chunk)
            idx = np.arange(s + offset, e + offset, dtype=np.uint64)
            base = (idx * np.uint64(3) + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)) & _M64
            acc = np.zeros(e - s, dtype=np.int64)
            for j in range(3):
                h = _splitmix64(base + np.uint64(j))
                for sh in (0, 16, 32, 48):
                    acc += ((h >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
            out[s:e] = (acc.astype(np.float64) - mean) * inv_std
    return out


def normal(shape, dtype: torch.dtype = torch.float16, seed: int = 0, std: float = 1.0,
           device="cpu") -> torch.Tensor:
    """Tensor of ~N(0, std²) values rounded once (RNE) from float64 to `dtype`."""
    numel = 1
    for s in shape:
        numel *= int(s)
    v = normal_f64(numel, seed)
    if std != 1.0:
        v = v * float(std)
    t = torch.from_numpy(v).to(dtype).reshape(tuple(shape))
    return t.to(device) if str(device) != "cpu" else t


def _splitmix64_t(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around = arithmetic mod 2^64; logical shifts by masking)."""
    def lsr(v, n):
        return (v >> n) & ((1 << (64 - n)) - 1)
    x = x + (-7046029254386353131)                  # 0x9E3779B97F4A7C15 as int64
    z = x
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)   # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def normal_device(shape, dtype: torch.dtype = torch.float16, seed: int = 0, std: float = 1.0, device="cuda") -> torch.Tensor:
    """`normal(shape, dtype, seed, std)` computed with torch integer ops ON `device`: the same hash, the same Irwin-Hall sum, the
    same float64 scale and the same single rounding to `dtype` -- bit-identical to the numpy form (tests/test_host_logic.py on
    the CPU, tests/test_gpu_parity.py on the GPU) at a few ms per 16 M values instead of seconds (bench.py's inputs)."""
    numel = 1
    for s in shape:
        numel *= int(s)
    dev = torch.device(device)
    out = torch.empty(numel, dtype=dtype, device=dev)
    inv_std = float(1.0 / np.sqrt(65536.0 * 65536.0 - 1.0))
    mean = 12 * 32767.5
    sm = (int(seed) * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF
    sm = sm - (1 << 64) if sm >= (1 << 63) else sm
    chunk = 1 << 24
    for s0 in range(0, numel, chunk):
        e = min(numel, s0 + chunk)
        idx = torch.arange(s0, e, dtype=torch.int64, device=dev)
        base = idx * 3 + sm
        acc = torch.zeros(e - s0, dtype=torch.int64, device=dev)
        for j in range(3):
            h = _splitmix64_t(base + j)
            for sh in (0, 16, 32, 48):
                acc += (h >> sh) & 0xFFFF
        v = (acc.to(torch.float64) - mean) * inv_std
        if std != 1.0:
            v = v * float(std)
        out[s0:e] = v.to(dtype)
    return out.reshape(tuple(shape))


def uniform_u64(numel: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64)
        return _splitmix64((idx + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)) & _M64)


def int8_tensor(shape, seed: int = 0) -> torch.Tensor:
    """Uniform int8 in [-127, 127]."""
    numel = 1
    for s in shape:
        numel *= int(s)
    h = uniform_u64(numel, seed)
    v = (h % np.uint64(255)).astype(np.int64) - 127
    return torch.from_numpy(v.astype(np.int8)).reshape(tuple(shape))
