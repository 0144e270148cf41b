"""The contract between the model stack and an attention implementation.

This is the interface the reference's model runner and graph runner program against
(mini_sglang/layers/attn/attn_backend.py:11-96); an object with these methods can be handed to the reference as
``model_runner.attn_backend`` and, the other way round, the runtime in this repository accepts any object that
honours it (the CPU oracle used by the tests is one).  Method names and argument names are part of the contract.

Call sequence
-------------
eager forward      ``init_forward_metadata(batch)`` once, then per layer ``forward(q, k, v, layer, batch)``
graph capture      ``init_cuda_graph_state(max_bs)`` once; per captured batch size
                   ``init_forward_metadata_capture_cuda_graph(bs, mode)`` and the layers' ``forward`` under capture
graph replay       ``init_forward_metadata_replay_cuda_graph(bs, req_pool_indices, seq_lens, seq_lens_cpu, mode)``
                   and then the replay itself

Tensors
-------
``q`` ``[T, Hq_local * D]`` or ``[T, Hq_local, D]`` (possibly a row-strided view of the qkv projection),
``k`` / ``v`` ``[T, Hkv_local, D]``; the result is a new ``[T, Hq_local * D]`` tensor.  ``layer`` carries
``num_heads, num_kv_heads, head_dim, scale, layer_id``.  With ``save_kv_cache`` the implementation stores k / v at
``batch.out_cache_loc`` BEFORE attending, so a decode step sees the token it has just produced.

Errors: a hook an implementation does not provide raises ``NotImplementedError`` naming the hook, exactly what a
caller probing for graph support expects.
"""
from __future__ import annotations

from abc import ABC, abstractmethod


class AttentionBackend(ABC):
    # ------------------------------------------------------------------ per forward
    @abstractmethod
    def init_forward_metadata(self, batch) -> None:
        """Derive whatever the layers' calls need from the batch (lengths, split sizes, prefix sums)."""

    def forward(self, q, k, v, layer, batch, save_kv_cache: bool = True, **kwargs):
        """Route one layer's call by the batch's forward mode."""
        run = self.forward_decode if batch.forward_mode.is_decode() else self.forward_extend
        return run(q, k, v, layer, batch, save_kv_cache=save_kv_cache, **kwargs)

    def forward_decode(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        """One query token per request against its whole cached context."""
        raise self._not_provided("forward_decode")

    def forward_extend(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        """Runs of new tokens per request, causal among themselves, on top of a cached prefix."""
        raise self._not_provided("forward_extend")

    # ------------------------------------------------------------------ graph capture and replay
    def init_cuda_graph_state(self, max_bs: int):
        """Allocate everything a captured decode step may touch, for batch sizes up to max_bs."""
        raise self._not_provided("init_cuda_graph_state")

    def init_forward_metadata_capture_cuda_graph(self, bs: int, forward_mode):
        """Point the metadata at static buffers before the step for batch size bs is captured."""
        raise self._not_provided("init_forward_metadata_capture_cuda_graph")

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices, seq_lens, seq_lens_cpu, forward_mode):
        """Refresh the static state in place for the batch about to be replayed."""
        raise self._not_provided("init_forward_metadata_replay_cuda_graph")

    def _not_provided(self, hook: str) -> NotImplementedError:
        return NotImplementedError(f"{type(self).__name__} does not implement {hook}")
