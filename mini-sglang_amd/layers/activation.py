"""SiluAndMul with the reference's module signature (mini_sglang/layers/activation.py:6-14)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class SiluAndMul(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        return ops.silu_and_mul(x2).view(*x.shape[:-1], x.shape[-1] // 2)
