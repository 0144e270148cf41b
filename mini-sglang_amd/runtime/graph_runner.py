"""hipGraph decode runner: one captured graph per batch size over static input buffers, replayed with padding.

Same contract as the reference's CudaGraphRunner (mini_sglang/managers/cuda_graph_runner.py:26-174): batch
sizes [1, 2, 4, 8, 16, 32, ...] (72-82), static ``input_ids / positions / out_cache_loc / seq_lens /
req_pool_indices`` buffers (94-98), the backend's three graph hooks (53-55, 105, 156-162), padded rows that
write the dummy slot 0 and attend to one token (142-151).  ``torch.cuda.CUDAGraph`` is hipGraph on ROCm.
Unlike the reference, padding also resets ``req_pool_indices`` of the padded rows to row 0, so a stale row
index can never point the kernel at a table row that a later request owns.
"""
from __future__ import annotations

import bisect
from typing import Dict, List

import torch

from .batch_info import BatchInfo, ForwardMode


def capture_batch_sizes(max_bs: int) -> List[int]:
    if max_bs < 4:
        return [1, 2, 3][:max_bs]
    if max_bs < 8:
        return [1, 2, 4] + ([max_bs] if max_bs != 4 else [])
    return [1, 2, 4, 8] + list(range(16, max_bs + 1, 16))


class HipGraphRunner:
    def __init__(self, model_runner, batch_sizes=None):
        self.mr = model_runner
        args = model_runner.server_args
        self.max_bs = min(args.max_num_reqs, args.max_capture_bs)
        self.graph_bs = sorted(batch_sizes) if batch_sizes else capture_batch_sizes(self.max_bs)
        self.max_bs = self.graph_bs[-1]
        dev = model_runner.device
        self.input_ids = torch.zeros(self.max_bs, dtype=torch.int32, device=dev)
        self.positions = torch.zeros(self.max_bs, dtype=torch.int32, device=dev)
        self.out_cache_loc = torch.zeros(self.max_bs, dtype=torch.int32, device=dev)
        self.seq_lens = torch.ones(self.max_bs, dtype=torch.int32, device=dev)
        self.req_pool_indices = torch.zeros(self.max_bs, dtype=torch.int32, device=dev)
        # keyed by batch size; a second set, captured on first use, also contains the backend's shared-prefix kernel
        self.graphs: Dict[int, torch.cuda.CUDAGraph] = {}
        self.outputs: Dict[int, torch.Tensor] = {}
        self.graphs_shared: Dict[int, torch.cuda.CUDAGraph] = {}
        self.outputs_shared: Dict[int, torch.Tensor] = {}
        self.pool = None
        self.mr.attn_backend.init_cuda_graph_state(self.max_bs)
        with torch.inference_mode():
            for bs in reversed(self.graph_bs):
                self._capture(bs)

    def _static_batch(self, bs: int) -> BatchInfo:
        mr = self.mr
        return BatchInfo(forward_mode=ForwardMode.DECODE, input_ids=self.input_ids[:bs], positions=self.positions[:bs],
                         seq_lens=self.seq_lens[:bs], out_cache_loc=self.out_cache_loc[:bs],
                         req_pool_indices=self.req_pool_indices[:bs], req_to_token_pool=mr.req_to_token_pool,
                         kv_cache_pool=mr.kv_cache_pool, page_allocator=mr.page_allocator, attn_backend=mr.attn_backend,
                         reqs=[], device=mr.device)

    @torch.inference_mode()
    def _capture(self, bs: int, shared_prefix: bool = False) -> None:
        batch = self._static_batch(bs)
        be = self.mr.attn_backend
        if shared_prefix:  # captured in the middle of serving: the static inputs go back to a harmless batch first
            self.input_ids.zero_(), self.positions.zero_(), self.out_cache_loc.zero_(), self.req_pool_indices.zero_()
            self.seq_lens.fill_(1)
            be.capture_shared_prefix = True
        try:
            be.init_forward_metadata_capture_cuda_graph(bs, batch.forward_mode)
        finally:
            if shared_prefix:
                be.capture_shared_prefix = False
        for _ in range(2):  # warm-up outside capture (library handles, lazy init)
            torch.cuda.synchronize()
            self.mr.model(batch.input_ids, batch.positions, batch)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            out = self.mr.model(batch.input_ids, batch.positions, batch)
        self.pool = g.pool()
        (self.graphs_shared if shared_prefix else self.graphs)[bs] = g
        (self.outputs_shared if shared_prefix else self.outputs)[bs] = out
        torch.cuda.synchronize()

    def can_run(self, batch: BatchInfo) -> bool:
        return batch.forward_mode.is_decode() and batch.batch_size <= self.graph_bs[-1]

    def replay(self, batch: BatchInfo) -> torch.Tensor:
        raw = batch.batch_size
        bs = self.graph_bs[bisect.bisect_left(self.graph_bs, raw)]
        if bs != raw:
            self.seq_lens.fill_(1)
            self.out_cache_loc.zero_()
            self.req_pool_indices.zero_()
        be = self.mr.attn_backend
        shared = hasattr(be, "set_graph_shared_groups") and be.set_graph_shared_groups(batch.shared_groups(), bs)
        if shared and bs not in self.graphs_shared:
            self._capture(bs, shared_prefix=True)
            be.set_graph_shared_groups(batch.shared_groups(), bs)
        self.input_ids[:raw].copy_(batch.input_ids)
        self.positions[:raw].copy_(batch.positions)
        self.seq_lens[:raw].copy_(batch.seq_lens)
        self.out_cache_loc[:raw].copy_(batch.out_cache_loc)
        self.req_pool_indices[:raw].copy_(batch.req_pool_indices)
        if hasattr(self.mr.attn_backend, "set_graph_reuse_lens"):
            self.mr.attn_backend.set_graph_reuse_lens(batch.shared_prefix_lens)
        self.mr.attn_backend.init_forward_metadata_replay_cuda_graph(
            bs, req_pool_indices=self.req_pool_indices, seq_lens=self.seq_lens, seq_lens_cpu=batch.seq_lens_cpu,
            forward_mode=batch.forward_mode)
        (self.graphs_shared if shared else self.graphs)[bs].replay()
        return (self.outputs_shared if shared else self.outputs)[bs][:raw]
