"""RMSNorm with the reference's module signature (mini_sglang/layers/layernorm.py:20-50)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn

from .. import ops


class RMSNorm(nn.Module):
    """``forward(x)`` returns a new tensor; ``forward(x, residual)`` updates BOTH tensors in place
    (residual <- x + residual, x <- norm of the unrounded sum) and returns ``(x, residual)`` — the
    in-place form the reference uses with its fused kernel (layernorm.py:33-35), with the rounding
    points of its torch path (layernorm.py:40-46)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.hidden_size = hidden_size
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        shape = x.shape
        if residual is not None:
            if not (x.is_contiguous() and residual.is_contiguous()):
                raise ValueError("fused add RMSNorm works in place and needs contiguous x and residual")
            ops.fused_add_rmsnorm(x.view(-1, self.hidden_size), residual.view(-1, self.hidden_size), self.weight.data, self.eps)
            return x, residual
        x2 = x if x.dim() == 2 else x.reshape(-1, self.hidden_size)
        return ops.rmsnorm(x2, self.weight.data, self.eps).view(shape)
