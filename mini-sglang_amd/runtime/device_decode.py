"""Decode steps chained on the GPU: batch preparation, forward and greedy sampling of step n+1 follow step n
without a host round trip, all inside one captured hipGraph per batch size.

The reference prepares every decode step on the host (mini_sglang/managers/batch_info.py:247-302), replays the
forward, samples, and reads the tokens back (scheduler.py:409-419) before it can prepare the next step.  Here
``msgl_prepare_decode`` does the preparation on the device with the allocator's own page order, so ``run(k)``
launches k graph replays back to back and the host synchronises once, in ``sync_host()``, which brings the
host-side state (requests' tokens and last slots, ``seq_lens_cpu``, the page allocator's free list) to exactly
what k host-prepared steps would have left behind.  Greedy sampling only (temperature 0).
"""
from __future__ import annotations

from typing import List

import torch

from .. import ops
from .batch_info import BatchInfo, ForwardMode


class DeviceDecodeLoop:
    def __init__(self, model_runner, batch: BatchInfo, max_steps: int = 64):
        mr = model_runner
        assert batch.forward_mode is not None and all(r.sampling_params.temperature == 0 for r in batch.reqs), \
            "device-chained decode samples greedily"
        self.mr, self.batch, self.max_steps = mr, batch, max_steps
        dev = mr.device
        bs = batch.batch_size
        self.bs = bs
        i32 = dict(dtype=torch.int32, device=dev)
        # static buffers the captured graph reads and writes
        self.req_pool_indices = batch.req_pool_indices.to(**i32).clone()
        self.seq_lens = batch.seq_lens.to(**i32).clone()
        self.last_slots = torch.tensor([r.last_slot for r in batch.reqs], **i32)
        self.next_tokens = torch.tensor([r.last_token_id for r in batch.reqs], **i32)
        self.input_ids = torch.zeros(bs, **i32)
        self.positions = torch.zeros(bs, **i32)
        self.out_cache_loc = torch.zeros(bs, **i32)
        # device copy of the allocator's free list: fixed capacity (every page of the pool), valid range in free_range
        self.free_pages = torch.zeros(mr.page_allocator.page_num, **i32)
        self.free_range = torch.zeros(2, **i32)
        self._upload_free_list()
        self.status = torch.zeros(1, **i32)
        self.token_log = torch.zeros((max_steps, bs), **i32)
        self.step_idx = torch.zeros(1, dtype=torch.int64, device=dev)
        self.pending = 0          # steps launched since the last sync_host()
        # head GEMM and greedy reduction in one kernel (no logits in memory) where the kernel is built for the shape;
        # its scratch is allocated here, outside the captured step
        head = mr.model.lm_head
        # (up to 128 requests the product is bound by the pass over the head's rows and the fused kernel is ahead of
        # GEMM + argmax; beyond, it would take two passes over the rows and the library GEMM wins)
        self.fused_head = bool(head.is_cuda and head.dtype == torch.bfloat16 and head.shape[1] % 256 == 0 and bs <= 128
                               and getattr(mr, "fused_head", True))
        self.head_ws = ops.lm_head_argmax_workspace(bs, head.shape[0], dev) if self.fused_head else None
        self.seq_lens_cpu = batch.seq_lens_cpu.clone()
        self.static_batch = BatchInfo(
            forward_mode=ForwardMode.DECODE, input_ids=self.input_ids, positions=self.positions, seq_lens=self.seq_lens,
            out_cache_loc=self.out_cache_loc, req_pool_indices=self.req_pool_indices, req_to_token_pool=mr.req_to_token_pool,
            kv_cache_pool=mr.kv_cache_pool, page_allocator=mr.page_allocator, attn_backend=mr.attn_backend, reqs=[], device=dev)
        be = mr.attn_backend
        if getattr(be, "decode_graph_vars", None) is None:
            be.init_cuda_graph_state(max(bs, getattr(mr.graph_runner, "max_bs", 0) if mr.graph_runner else 0))
        # runs of requests on a shared radix prefix: this loop's graph then contains the backend's shared-prefix kernel
        self.shared = hasattr(be, "set_graph_shared_groups") and be.set_graph_shared_groups(batch.shared_groups(), bs)
        have = getattr(be, "decode_graph_metadata_shared" if self.shared else "decode_graph_metadata", {})
        if bs not in have:
            be.capture_shared_prefix = self.shared
            try:
                be.init_forward_metadata_capture_cuda_graph(bs, ForwardMode.DECODE)
            finally:
                be.capture_shared_prefix = False
        self.graph = None
        self._capture()

    def _upload_free_list(self):
        """Bring the device free list in line with the host allocator (after host-side allocs / frees)."""
        fp = self.mr.page_allocator.free_page
        self.free_pages[: fp.numel()].copy_(fp)
        self.free_range.copy_(torch.tensor([0, fp.numel()], dtype=torch.int32))
        self._free_uploaded = fp.clone()

    def _head_argmax(self, hidden, head_rows):
        return ops.lm_head_argmax(hidden, head_rows, workspace=self.head_ws)

    # one step: prepare -> forward -> argmax -> log
    def _step(self):
        mr = self.mr
        ops.prepare_decode(self.req_pool_indices, self.seq_lens, self.last_slots, self.free_pages, self.free_range,
                           mr.req_to_token_pool.req_to_token, self.positions, self.out_cache_loc, self.status, mr.page_size,
                           next_tokens=self.next_tokens, input_ids=self.input_ids,
                           req_to_page=mr.req_to_token_pool.req_to_page if mr.page_size > 1 else None)
        if hasattr(mr.attn_backend, "refresh_decode_plan"):  # launch order for the lengths this step attends over
            mr.attn_backend.refresh_decode_plan(self.bs, self.seq_lens)
        # greedy sampling on the vocabulary shard of each rank; the ranks exchange (value, index) pairs, not logits
        mr.model.forward_greedy(self.input_ids, self.positions, self.static_batch,
                                lambda lg: ops.argmax(lg, return_values=True), out=self.next_tokens,
                                head_argmax_fn=self._head_argmax if self.fused_head else None)
        self.token_log.index_copy_(0, self.step_idx, self.next_tokens.view(1, -1))
        self.step_idx.add_(1)

    def _set_metadata(self, steps: int):
        """The split size is the only launch parameter that depends on the lengths; sized for the longest
        request after `steps` more tokens."""
        grown = self.seq_lens_cpu + steps
        if hasattr(self.mr.attn_backend, "set_graph_shared_groups"):
            self.mr.attn_backend.set_graph_shared_groups(self.batch.shared_groups() if self.shared else [], self.bs)
        self.mr.attn_backend.init_forward_metadata_replay_cuda_graph(
            self.bs, req_pool_indices=self.req_pool_indices, seq_lens=self.seq_lens, seq_lens_cpu=grown,
            forward_mode=ForwardMode.DECODE)

    @torch.inference_mode()
    def _capture(self):
        # warm-up on a snapshot of the mutable state, restored afterwards: capture must not consume pages or tokens
        keep = [t.clone() for t in (self.seq_lens, self.last_slots, self.next_tokens, self.free_range, self.step_idx)]
        # (the page-table entries and K/V slots the warm-up writes lie beyond the requests' lengths: never read)
        self._set_metadata(2)
        torch.cuda.synchronize()
        self._step()
        torch.cuda.synchronize()
        self._restore(keep)
        g = torch.cuda.CUDAGraph()
        pool = self.mr.graph_runner.pool if self.mr.graph_runner is not None else None
        with torch.cuda.graph(g, pool=pool):
            self._step()
        self.graph = g
        torch.cuda.synchronize()
        self._restore(keep)

    def _restore(self, keep):
        for t, k in zip((self.seq_lens, self.last_slots, self.next_tokens, self.free_range, self.step_idx), keep):
            t.copy_(k)
        self.status.zero_()
        torch.cuda.synchronize()

    @torch.inference_mode()
    def run(self, steps: int) -> None:
        """Launch `steps` chained decode steps (no synchronisation)."""
        assert self.pending + steps <= self.max_steps, "sync_host() first: the token log is full"
        free_now = self.mr.page_allocator.free_page.numel()
        worst = self.bs * (-(-(self.pending + steps) // self.mr.page_size) + 1) if self.mr.page_size > 1 else self.bs * (self.pending + steps)
        if worst > free_now:
            raise RuntimeError(f"Not enough free pages available. need up to {worst}, free {free_now}")
        if self.pending == 0 and not torch.equal(self._free_uploaded, self.mr.page_allocator.free_page):
            self._upload_free_list()  # the host allocated or freed pages since (another batch's prefill, an eviction)
        self._set_metadata(self.pending + steps)
        for _ in range(steps):
            self.graph.replay()
        self.pending += steps

    def sync_host(self) -> List[List[int]]:
        """Wait for the launched steps and fold them into the host state.  Returns the new tokens per request."""
        k = self.pending
        if k == 0:
            return [[] for _ in range(self.bs)]
        torch.cuda.synchronize()
        status = int(self.status.item())
        # a refused step changes nothing and every later one is refused too: the accepted steps are a prefix
        done = int(self.seq_lens[0].item()) - int(self.seq_lens_cpu[0]) if status != 0 else k
        toks = self.token_log[:done].cpu().t().tolist() if done else [[] for _ in range(self.bs)]
        last = self.last_slots.cpu().tolist()
        b = self.batch
        for r, new, s in zip(b.reqs, toks, last):
            r.token_ids.extend(int(t) for t in new)
            r.last_slot = int(s)
        self.seq_lens_cpu = self.seq_lens.cpu()
        b.seq_lens_cpu = self.seq_lens_cpu.clone()
        b.seq_lens = self.seq_lens.clone()
        b.forward_mode = ForwardMode.DECODE
        taken = int(self.free_range[0].item())
        self.mr.page_allocator.alloc(taken)          # the same pages, in the same order, leave the host free list
        self.step_idx.zero_()
        self.pending = 0
        self._upload_free_list()                     # positions restart at 0 of what the host list now holds
        if status != 0:
            # the refused steps' forward passes left garbage there: the last accepted token is the next input again
            self.next_tokens.copy_(torch.tensor([r.token_ids[-1] for r in b.reqs], dtype=torch.int32))
            self.status.zero_()
            raise RuntimeError(f"device decode step refused after {done} of {k} steps: "
                               + ("free list exhausted" if status == 1 else "context longer than the page table row"))
        return toks
