"""Attention layer shim (mini_sglang/layers/attention.py:11-33): holds the head geometry and hands
q, k, v to the batch's attention backend."""
from __future__ import annotations

import torch
from torch import nn


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scale: float, num_kv_heads: int, layer_id: int):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale
        self.num_kv_heads = num_kv_heads
        self.layer_id = layer_id

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch, save_kv_cache: bool = True) -> torch.Tensor:
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        return batch.attn_backend.forward(q=q, k=k, v=v, layer=self, batch=batch, save_kv_cache=save_kv_cache)
