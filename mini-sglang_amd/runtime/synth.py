"""Deterministic synthetic data for tests, fixtures and benchmarks.

Values come from a counter-based integer hash (splitmix64) evaluated with wrapping int64 tensor
arithmetic and turned into an approximately normal variate by summing four 16-bit uniforms
(Irwin-Hall), so the same (seed, index) gives the same bits on every machine, device and library
version — no dependence on a framework RNG stream.
"""
from __future__ import annotations

from typing import Sequence

import torch

_C0 = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z: torch.Tensor, n: int) -> torch.Tensor:
    return (z >> n) & ((1 << (64 - n)) - 1)


def hash64(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """splitmix64 of (idx + (seed+1) * golden-ratio); int64 in, int64 (raw 64 bits) out."""
    z = idx + _s64((seed + 1) * _C0)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    return z ^ _lsr(z, 31)


def normal(shape: Sequence[int], seed: int, std: float = 1.0, dtype=torch.bfloat16, device="cpu",
           chunk: int = 1 << 24) -> torch.Tensor:
    """Approximately N(0, std^2) tensor (support +-3.46 std), identical for a given (shape, seed)."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=dtype, device=device)
    for start in range(0, n, chunk):
        stop = min(n, start + chunk)
        h = hash64(torch.arange(start, stop, dtype=torch.int64, device=device), seed)
        s = (h & 0xFFFF) + (_lsr(h, 16) & 0xFFFF) + (_lsr(h, 32) & 0xFFFF) + (_lsr(h, 48) & 0xFFFF)
        x = (s - 131070).to(torch.float32) / 37837.0  # Irwin-Hall(4) centred and scaled to unit variance
        out[start:stop] = (x * std).to(dtype)
    return out.reshape(tuple(shape))


def uniform_int(n: int, seed: int, low: int, high: int, device="cpu") -> torch.Tensor:
    """n integers in [low, high) (int64)."""
    h = hash64(torch.arange(n, dtype=torch.int64, device=device), seed)
    return low + (_lsr(h, 1) % (high - low))


def permutation(n: int, seed: int, device="cpu") -> torch.Tensor:
    """A fixed pseudo-random permutation of range(n): argsort of hashes (ties impossible for distinct idx
    in practice; stable sort keeps it deterministic anyway)."""
    h = hash64(torch.arange(n, dtype=torch.int64, device=device), seed)
    return torch.sort(_lsr(h, 1), stable=True).indices
