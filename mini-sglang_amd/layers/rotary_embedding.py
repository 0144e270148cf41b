"""Neox rotary embedding with the reference's module signature
(mini_sglang/layers/rotary_embedding.py:36-106)."""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import torch
from torch import nn

from .. import ops


def build_cos_sin_cache(head_size: int, max_position: int, base: float) -> torch.Tensor:
    """fp32 [max_position, head_size] = cat(cos(t * inv_freq), sin(t * inv_freq)); computed on the host in
    fp32 exactly as the reference does (rotary_embedding.py:59-67) so the table is bit-identical."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float, device="cpu") / head_size))
    t = torch.arange(max_position, dtype=torch.float, device="cpu")
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


class RotaryEmbedding(nn.Module):
    """``forward(positions, query, key)`` rotates query [T, Hq*D] and key [T, Hkv*D] IN PLACE (views of the
    qkv GEMM output are fine) and returns them.  ``reference_rounding=True`` reproduces the reference's
    torch path bit for bit; False keeps fp32 inside the kernel (its sgl_kernel path)."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 reference_rounding: bool = True) -> None:
        super().__init__()
        if rotary_dim != head_size:
            raise ValueError("partial rotary is not supported (rotary_dim must equal head_size)")
        self.head_size = head_size
        self.reference_rounding = reference_rounding
        self.register_buffer("cos_sin_cache", build_cos_sin_cache(head_size, max_position_embeddings, base), persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        ops.rope_neox_(positions, query, key, self.cos_sin_cache, self.head_size, self.reference_rounding)
        return query, key


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float) -> RotaryEmbedding:
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
