"""The attention-backend interface the model runner and the graph runner call
(mini_sglang/layers/attn/attn_backend.py:11-96): same method names, arguments and error behaviour."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch


class AttentionBackend(ABC):
    @abstractmethod
    def init_forward_metadata(self, batch):
        """Called once per eager forward, before the first layer."""
        raise NotImplementedError()

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, layer, batch, save_kv_cache: bool = True, **kwargs):
        if batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, batch, save_kv_cache=save_kv_cache, **kwargs)
        return self.forward_extend(q, k, v, layer, batch, save_kv_cache=save_kv_cache, **kwargs)

    def forward_decode(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def init_cuda_graph_state(self, max_bs: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs: int, forward_mode):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                                                seq_lens_cpu: torch.Tensor, forward_mode):
        raise NotImplementedError()
