"""The per-layer attention module: head geometry plus a hand-off to whichever backend the batch carries.

Same constructor arguments and attributes as the reference's layer (mini_sglang/layers/attention.py:11-33), because
backends read them (``layer.num_heads``, ``layer.num_kv_heads``, ``layer.head_dim``, ``layer.scale``,
``layer.layer_id``).  ``forward`` additionally accepts ``save_kv_cache=False`` for callers that have already written
k / v into the pool (the fused q/k-norm + RoPE + store kernel does).
"""
from __future__ import annotations

from torch import nn


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scale: float, num_kv_heads: int, layer_id: int):
        super().__init__()
        if num_heads % num_kv_heads:
            raise ValueError(f"{num_heads} query heads cannot be grouped over {num_kv_heads} kv heads")
        self.layer_id = layer_id
        self.num_heads, self.num_kv_heads = num_heads, num_kv_heads
        self.head_dim, self.scale = head_dim, scale

    def extra_repr(self) -> str:
        return f"layer {self.layer_id}: {self.num_heads} q / {self.num_kv_heads} kv heads x {self.head_dim}, scale {self.scale:.4g}"

    def forward(self, q, k, v, batch, save_kv_cache: bool = True):
        kv_shape = (-1, self.num_kv_heads, self.head_dim)   # views: v is a strided slice of the qkv projection
        return batch.attn_backend.forward(q=q, k=k.view(kv_shape), v=v.view(kv_shape), layer=self, batch=batch,
                                          save_kv_cache=save_kv_cache)
