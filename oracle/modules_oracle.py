"""Oracle-backed drop-ins for the layer namespace and the attention backend (TEST INFRASTRUCTURE ONLY).

They let tests run the same Qwen3-shaped stack once through the HIP kernels and once through the CPU
restatement of the reference's torch path (oracle/torch_oracle.py) on identical weights and page tables.
The op sequence per layer is the reference's unfused one (mini_sglang/models/qwen3.py:88-95): there is no
``qknorm_rope_store`` here, so the model falls back to module-per-op calls.
"""
from __future__ import annotations

from functools import lru_cache

import torch
from torch import nn

from . import torch_oracle as orc


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def forward(self, x, residual=None):
        return orc.rmsnorm_ref(x, self.weight, self.eps, residual)


class SiluAndMul(nn.Module):
    def forward(self, x):
        return orc.silu_and_mul_ref(x)


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size, rotary_dim, max_position_embeddings, base):
        super().__init__()
        assert rotary_dim == head_size
        self.head_size = head_size
        self.reference_rounding = True
        self.register_buffer("cos_sin_cache", orc.build_cos_sin_cache(head_size, max_position_embeddings, base), persistent=False)

    def forward(self, positions, query, key):
        return orc.rope_ref(positions, query, key, self.cos_sin_cache, self.head_size)


@lru_cache(1)
def get_rope(head_size, rotary_dim, max_position, base):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)


class Attention(nn.Module):
    def __init__(self, num_heads, head_dim, scale, num_kv_heads, layer_id):
        super().__init__()
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads, self.layer_id = num_heads, head_dim, scale, num_kv_heads, layer_id

    def forward(self, q, k, v, batch, save_kv_cache=True):
        k = k.reshape(-1, self.num_kv_heads, self.head_dim)
        v = v.reshape(-1, self.num_kv_heads, self.head_dim)
        return batch.attn_backend.forward(q=q, k=k, v=v, layer=self, batch=batch, save_kv_cache=save_kv_cache)


class OracleAttnBackend:
    """The reference's TorchNativeAttnBackend (mini_sglang/layers/attn/torch_attn_backend.py:14-252) over the
    oracle functions; implements no graph hooks, like the original."""

    def __init__(self, model_runner):
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.kv_cache_pool = model_runner.kv_cache_pool

    def init_forward_metadata(self, batch):
        pass

    def forward(self, q, k, v, layer, batch, save_kv_cache=True, **kw):
        k_cache, v_cache = self.kv_cache_pool.get_kv_buffer(layer.layer_id)
        if save_kv_cache:
            orc.store_kv_ref(k_cache, v_cache, batch.out_cache_loc, k, v)
        q3 = q.reshape(-1, layer.num_heads, layer.head_dim)
        r2t = self.req_to_token_pool.req_to_token
        if batch.forward_mode.is_decode():
            o = orc.paged_decode_ref(q3, k_cache, v_cache, r2t, batch.req_pool_indices, batch.seq_lens, layer.scale)
        else:
            o = orc.paged_extend_ref(q3, k_cache, v_cache, r2t, batch.req_pool_indices, batch.seq_lens, batch.prefix_seq_lens,
                                     batch.input_seq_lens, layer.scale)
        return o.reshape(q.shape[0], layer.num_heads * layer.head_dim)
