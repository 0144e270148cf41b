"""Gated SiLU as a module with the reference's signature (mini_sglang/layers/activation.py:6-14): the input's last
dimension holds gate and up projections side by side, the output is ``silu(gate) * up`` at half the width.  One HIP
kernel (msgl_silu_and_mul), one pass over the data; there is no torch fallback."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        width = x.shape[-1]
        if width % 2:
            raise ValueError(f"last dimension {width} cannot be split into gate and up halves")
        rows = x if x.dim() == 2 else x.reshape(-1, width)
        return ops.silu_and_mul(rows).view(*x.shape[:-1], width // 2)
