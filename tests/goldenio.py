"""Loader for the committed golden fixtures (tests/golden/*.npz, produced by make_golden.py)."""
import json
import os

import numpy as np
import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
NAME = {v: k for k, v in DT.items()}


def from_bits(arr: np.ndarray, dtype: torch.dtype = None) -> torch.Tensor:
    """Inverse of make_golden.bits(): integer bit patterns -> tensor of `dtype`."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype == np.uint16:
        return torch.from_numpy(arr.view(np.int16)).view(dtype)
    if arr.dtype == np.uint32:
        return torch.from_numpy(arr.view(np.int32)).view(torch.float32)
    return torch.from_numpy(arr)


def bits_equal(a: torch.Tensor, b: torch.Tensor) -> bool:
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype in (torch.float16, torch.bfloat16):
        return torch.equal(a.view(torch.int16), b.view(torch.int16))
    if a.dtype == torch.float32:
        return torch.equal(a.view(torch.int32), b.view(torch.int32))
    return torch.equal(a, b)


def n_mismatch(a: torch.Tensor, b: torch.Tensor) -> int:
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    if a.dtype in (torch.float16, torch.bfloat16):
        return int((a.view(torch.int16) != b.view(torch.int16)).sum())
    if a.dtype == torch.float32:
        return int((a.view(torch.int32) != b.view(torch.int32)).sum())
    return int((a != b).sum())


def rel_fro(y: torch.Tensor, ref: torch.Tensor) -> float:
    """Frobenius relative error (the parity metric of SURVEY.md §8d)."""
    y, ref = y.detach().cpu().double(), ref.detach().cpu().double()
    den = ref.norm().item()
    return (y - ref).norm().item() / (den if den > 0 else 1.0)


class Golden:
    def __init__(self):
        with open(os.path.join(HERE, "manifest.json")) as f:
            self.manifest = json.load(f)
        with open(os.path.join(HERE, "g3_digests.json")) as f:
            self.g3 = json.load(f)
        self._npz = {}

    def npz(self, name):
        if name not in self._npz:
            self._npz[name] = np.load(os.path.join(HERE, name))
        return self._npz[name]
