"""``--attention_backend hip``: paged attention on hand-written gfx950 kernels.

Plugs in where the reference selects a backend by name (mini_sglang/managers/model_runner.py:107-128)
and is constructed the same way, ``HipAttnBackend(model_runner)``, reading ``device``,
``req_to_token_pool``, ``kv_cache_pool``, ``page_size`` and ``model_config.max_context_len`` from it
(compare mini_sglang/layers/attn/fa3_attn_backend.py:34-41).

Differences from the FA3 backend that matter to a caller:
  * no page-table copy: the kernels read ``req_to_token`` + ``req_pool_indices`` + ``seq_lens``
    directly, so the per-forward metadata is two integers (decode) or one prefix sum (extend), and the
    graph-replay hook only refreshes the split size in a 4-byte device word;
  * page ids are never read from ``req_to_page`` (its writer is wrong for multi-page prefixes,
    mini_sglang/mem_cache/req2token.py:86-91); ``req_to_token`` is the only table used;
  * q may be a row-strided view of the qkv GEMM output (no ``.contiguous()`` copy);
  * scheduling metadata is built on the device once per forward and shared by all layers: the launch order of
    ragged decode and extend batches (longest work item first), and, for decode batches in which neighbouring
    requests sit on one radix prefix (``batch.shared_groups()``), the group arrays that let that prefix be
    attended once per group instead of once per request.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import ops
from .backend_api import AttentionBackend


@dataclass
class HipAttnMetadata:
    # decode
    max_splits: int = 1
    chunk_tokens: int = 0
    chunk_dev: Optional[torch.Tensor] = None
    reuse_lens: Optional[torch.Tensor] = None  # per request: leading positions shared with other requests
    decode_plan: Optional[torch.Tensor] = None  # launch order of the (split, request) items (ragged batches)
    shared: Optional[dict] = None  # groups of requests on one radix prefix, attended once per group
    # extend
    cu_seqlens_q: Optional[torch.Tensor] = None
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    extend_plan: Optional[torch.Tensor] = None  # launch order of the (request, query tile) items, longest first
    # metadata of a captured batch size: its launches use the workspace that was sized before the first capture
    static: bool = False


# work items of the shared-prefix kernel in a captured launch (times the kv heads = work groups; one round of the chip when idle)
GRAPH_SHARED_ITEMS = 32


def splits_for_batch(bs: int) -> int:
    """Upper bound on splits per request (the launch has bs * splits work groups, most of which exit at once
    for short requests, at about 2 ns each): up to 4096 groups in all, so a 32k-token request in a batch of 256
    can still be cut into 2k-token pieces, and a single 128k-token request into 512."""
    return max(1, min(512, -(-4096 // max(bs, 1))))


class HipAttnBackend(AttentionBackend):
    def __init__(self, model_runner):
        super().__init__()
        self.forward_metadata: Optional[HipAttnMetadata] = None
        self.device = model_runner.device
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.kv_cache_pool = model_runner.kv_cache_pool
        self.page_size = model_runner.page_size
        cfg = model_runner.model_config
        self.max_context_len = cfg.max_context_len
        # this rank's heads, as the reference shards them (mini_sglang/models/qwen3.py:42-46)
        tp_size = int(getattr(model_runner, "tp_size", 1) or 1)
        self.num_heads = cfg.num_heads // tp_size
        self.num_kv_heads = max(1, cfg.num_kv_heads // tp_size)
        self.head_dim = cfg.head_dim
        # Two split workspaces.  Captured launches have the address of theirs baked in, so that one is allocated
        # once, in init_cuda_graph_state, for the worst case over all batch sizes, and never again.  Eager launches
        # use a buffer of their own that grows on demand.
        self._graph_workspace: Optional[torch.Tensor] = None
        self._eager_workspace: Optional[torch.Tensor] = None
        self._extend_workspace: Optional[torch.Tensor] = None
        # K/V rows are read once per step: stream them past the caches (non-temporal), keeping only the
        # prefix positions that several requests share cacheable
        self.kv_stream = True
        # eager decode steps attend a radix prefix that neighbouring requests share once per run of such requests
        self.share_prefix_reads = True
        # ... and captured steps too: a runner captures a second graph per batch size that contains the shared-prefix
        # kernel (capture_shared_prefix = True while it does) and replays it for batches that announce such runs
        self.share_prefix_in_graphs = True
        self.capture_shared_prefix = False
        self._graph_groups = []          # runs of requests on a shared prefix in the batch about to be replayed
        self._graph_use_shared = False   # ... and whether that batch replays the graph with the shared-prefix kernel
        self._graph_shared_key = None    # what the static group buffers currently describe
        self.decode_graph_metadata_shared: Dict[int, HipAttnMetadata] = {}
        self._planned_split = (None, 0)
        self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count if torch.cuda.is_available() else 256
        self.decode_graph_vars: Optional[dict] = None
        self.decode_graph_metadata: Dict[int, HipAttnMetadata] = {}

    # ------------------------------------------------------------------ helpers
    def _workspace_for(self, md: HipAttnMetadata, bs: int, Hq: int, D: int) -> Optional[torch.Tensor]:
        if md.max_splits <= 1:
            return None
        need = ops.decode_workspace_bytes(bs, Hq, D, md.max_splits)
        if md.static:
            ws = self._graph_workspace
            if ws is None or ws.numel() < need:
                raise RuntimeError(f"captured decode step needs a {need}-byte split workspace but init_cuda_graph_state "
                                   f"sized {0 if ws is None else ws.numel()} bytes ({Hq} query heads here, {self.num_heads} announced)")
            return ws
        if self._eager_workspace is None or self._eager_workspace.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("an eager decode workspace cannot grow during graph capture; captured steps take their "
                                   "metadata from init_forward_metadata_capture_cuda_graph")
            self._eager_workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._eager_workspace

    def _extend_workspace_for(self, md: HipAttnMetadata, batch, T: int, layer) -> Optional[torch.Tensor]:
        """Room for key-split partials when the extend launch would not cover the chip (few 128-row query tiles x head
        pairs with long key ranges: a short prompt on a long cached prefix, few heads per rank); None otherwise.  Extend
        is never captured, so the buffer may grow between forwards."""
        ks = ops.extend_key_splits(batch.seq_lens.numel(), md.max_seqlen_q, md.max_seqlen_k, layer.num_heads, layer.num_kv_heads)
        if ks <= 1:
            return None
        # sized for the runs the library will take for this launch (not for the worst case of eight)
        need = ops.extend_workspace_bytes_for(T, layer.num_heads, layer.head_dim, ks)
        if self._extend_workspace is None or self._extend_workspace.numel() < need:
            self._extend_workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._extend_workspace

    def _decode_split(self, seq_lens_cpu: torch.Tensor, bs: int, max_splits: int):
        """(tokens per split, ragged).  A batch whose longest request is well above the average is launched in
        planned order, with the split size a host-side model of that schedule likes best; the model costs a few
        milliseconds, so its answer is kept until the batch has grown by 64 tokens per request or changed."""
        kv_heads = self.num_kv_heads
        longest, total = int(seq_lens_cpu.max()), int(seq_lens_cpu.sum())
        if bs == 1 or longest * bs <= 1.25 * total:
            return ops.pick_split(longest, total, bs, max_splits, kv_heads), False
        key = (bs, max_splits, total // (64 * bs), longest // 64)
        if self._planned_split[0] != key:
            self._planned_split = (key, ops.pick_split_planned(seq_lens_cpu.tolist(), max_splits, kv_heads, self._cus))
        chunk = max(self._planned_split[1], -(-(-(-longest // max_splits)) // 16) * 16)
        return chunk, True

    # ------------------------------------------------------------------ eager metadata
    def init_forward_metadata(self, batch):
        md = HipAttnMetadata()
        bs = batch.seq_lens.numel()
        if batch.forward_mode.is_decode():
            md.max_splits = splits_for_batch(bs)
            groups = batch.shared_groups() if self.share_prefix_reads and hasattr(batch, "shared_groups") else []
            if groups:
                self.forward_metadata = self._shared_prefix_metadata(batch, bs, groups)
                return
            md.chunk_tokens, ragged = self._decode_split(batch.seq_lens_cpu, bs, md.max_splits)
            # eager launches know the longest sequence: launch only the splits that can be active
            md.max_splits = max(1, min(md.max_splits, -(-int(batch.seq_lens_cpu.max()) // md.chunk_tokens)))
            md.reuse_lens = getattr(batch, "shared_prefix_lens", None)
            if ragged and md.max_splits > 1:
                md.decode_plan = ops.decode_plan(batch.seq_lens, md.max_splits, md.chunk_tokens)
        else:
            md.max_seqlen_q = int(batch.input_seq_lens_max)
            md.max_seqlen_k = int(batch.seq_lens_cpu.max())
            md.cu_seqlens_q = torch.nn.functional.pad(torch.cumsum(batch.input_seq_lens, dim=0, dtype=torch.int32), (1, 0))
            if bs > 1:  # a single request's tiles are already launched longest first
                md.extend_plan = ops.extend_plan(md.cu_seqlens_q, batch.prefix_seq_lens, md.max_seqlen_q)
        self.forward_metadata = md

    def _shared_prefix_metadata(self, batch, bs: int, groups) -> HipAttnMetadata:
        """Eager decode of a batch in which runs of requests sit on a common radix prefix: the prefix is attended
        once per run (ops.shared_prefix_decode), every request's own tail by the gather kernel in planned order,
        and one merge joins the two.  (Captured steps: _refresh_graph_shared.)"""
        i32 = dict(dtype=torch.int32, device=self.device)
        key_chunk = 1024
        start, extra = [0] * bs, [0] * bs
        for members, plen in groups:
            for i in members:
                start[i], extra[i] = plen, -(-plen // key_chunk)
        lens = batch.seq_lens_cpu.tolist()
        rest = [max(1, s - p) for s, p in zip(lens, start)]
        most_extra = max(extra)
        own_splits = max(1, splits_for_batch(bs) - most_extra)
        kv_heads = self.num_kv_heads
        key = ("shared", bs, own_splits, sum(rest) // (64 * bs), max(rest) // 64, tuple(groups))
        if self._planned_split[0] != key:  # the schedule model costs milliseconds: keep its answer while the batch barely changes
            self._planned_split = (key, ops.pick_split_planned(rest, own_splits, kv_heads, self._cus))
        chunk = max(self._planned_split[1], -(-(-(-max(rest) // own_splits)) // 16) * 16)
        md = HipAttnMetadata(chunk_tokens=chunk, max_splits=-(-max(rest) // chunk) + most_extra)
        md.reuse_lens = getattr(batch, "shared_prefix_lens", None)
        first, flat = _flatten_groups(groups)
        md.shared = dict(first=torch.tensor(first, **i32), count=torch.tensor([len(g[0]) for g in groups], **i32),
                         prefix=torch.tensor([g[1] for g in groups], **i32), members=torch.tensor(flat, **i32),
                         max_members=max(len(g[0]) for g in groups), max_prefix=max(g[1] for g in groups), key_chunk=key_chunk,
                         start=torch.tensor(start, **i32), extra=torch.tensor(extra, **i32))
        md.decode_plan = ops.decode_plan(batch.seq_lens, md.max_splits, chunk, start_lens=md.shared["start"])
        return md

    # ------------------------------------------------------------------ forward
    def forward_extend(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        if save_kv_cache:
            self.kv_cache_pool.set_kv_buffer(layer, batch.out_cache_loc, k, v)
        md = self.forward_metadata
        k_cache, v_cache = self.kv_cache_pool.get_kv_buffer(layer.layer_id)
        T = q.shape[0]
        o = ops.paged_extend(
            _as_heads(q, layer.num_heads, layer.head_dim), k_cache, v_cache, self.req_to_token_pool.req_to_token,
            batch.req_pool_indices, md.cu_seqlens_q, batch.prefix_seq_lens, md.max_seqlen_q, layer.scale, self.page_size,
            max_seqlen_k=md.max_seqlen_k, plan=md.extend_plan, workspace=self._extend_workspace_for(md, batch, T, layer))
        return o.view(T, layer.num_heads * layer.head_dim)

    def forward_decode(self, q, k, v, layer, batch, save_kv_cache: bool = True):
        if save_kv_cache:
            self.kv_cache_pool.set_kv_buffer(layer, batch.out_cache_loc, k, v)
        md = self.forward_metadata
        k_cache, v_cache = self.kv_cache_pool.get_kv_buffer(layer.layer_id)
        bs = q.shape[0]
        q3 = _as_heads(q, layer.num_heads, layer.head_dim)
        ws = self._workspace_for(md, bs, layer.num_heads, layer.head_dim)
        sh = md.shared
        if sh is not None:
            ops.shared_prefix_decode(q3, k_cache, v_cache, self.req_to_token_pool.req_to_token, batch.req_pool_indices, batch.seq_lens,
                                     sh["start"], sh["first"], sh["count"], sh["prefix"], sh["max_members"], sh["max_prefix"],
                                     sh["key_chunk"], layer.scale, ws, md.max_splits, md.chunk_tokens, chunk_dev=md.chunk_dev,
                                     items=sh.get("items"), key_chunk_dev=sh.get("key_chunk_dev"), group_members=sh["members"])
        o = ops.paged_decode(
            q3, k_cache, v_cache, self.req_to_token_pool.req_to_token,
            batch.req_pool_indices, batch.seq_lens, layer.scale, self.page_size, workspace=ws,
            max_splits=md.max_splits, chunk_tokens=md.chunk_tokens, chunk_dev=md.chunk_dev,
            kv_stream=self.kv_stream, reuse_lens=md.reuse_lens, plan=md.decode_plan,
            start_lens=sh["start"] if sh is not None else None, extra_splits=sh["extra"] if sh is not None else None)
        return o.view(bs, layer.num_heads * layer.head_dim)

    # ------------------------------------------------------------------ graph hooks
    def init_cuda_graph_state(self, max_bs: int):
        """Static state shared by all captured batch sizes (compare fa3_attn_backend.py:154-175): the 4-byte
        split-size word, the launch-order and shared-prefix buffers, and the split workspace, sized here once for the
        worst (bs, splits_for_batch(bs)) pair — bs * splits is not monotone in bs — from the head counts the
        constructor read off the runner's model config."""
        most_items = max(bs * splits_for_batch(bs) for bs in range(1, max_bs + 1))
        self.decode_graph_vars = {"chunk": torch.full((1,), 1 << 20, dtype=torch.int32, device=self.device),
                                  "reuse_lens": torch.zeros(max_bs, dtype=torch.int32, device=self.device),
                                  # launch order of the captured decode kernels and the word that says whether the
                                  # next refresh sorts it (ragged batch) or writes the identity (uniform batch)
                                  "plan": torch.arange(most_items, dtype=torch.int32, device=self.device),
                                  "plan_sorted": torch.zeros(1, dtype=torch.int32, device=self.device)}
        z = lambda n, v=0: torch.full((n,), v, dtype=torch.int32, device=self.device)
        # group arrays, per-request start / extra-split counts, the shared-prefix kernel's work list and slice length
        self.decode_graph_vars.update(sh_first=z(max_bs), sh_count=z(max_bs), sh_prefix=z(max_bs), sh_members=z(max_bs), sh_start=z(max_bs),
                                      sh_extra=z(max_bs), sh_items=z(GRAPH_SHARED_ITEMS, -1), sh_key_chunk=z(1, 1024))
        self._graph_shared_key = None
        self.decode_graph_metadata = {}
        self.decode_graph_metadata_shared = {}
        worst = max(ops.decode_workspace_bytes(bs, self.num_heads, self.head_dim, splits_for_batch(bs)) for bs in range(1, max_bs + 1))
        self._graph_workspace = torch.empty(max(worst, 16), dtype=torch.uint8, device=self.device)

    def init_forward_metadata_capture_cuda_graph(self, bs: int, forward_mode):
        if forward_mode.is_extend():
            raise NotImplementedError("extend is not captured in a graph")
        splits = splits_for_batch(bs)
        md = HipAttnMetadata(static=True, max_splits=splits, chunk_dev=self.decode_graph_vars["chunk"],
                             reuse_lens=self.decode_graph_vars["reuse_lens"][:bs],
                             decode_plan=self.decode_graph_vars["plan"][: bs * splits] if splits > 1 else None)
        if md.decode_plan is not None:  # valid for any lengths until the first refresh
            md.decode_plan.copy_(torch.arange(bs * splits, dtype=torch.int32, device=self.device))
        if self.capture_shared_prefix:
            assert self.graph_can_share_prefix(bs), "no room for shared-prefix partials at this batch size"
            v = self.decode_graph_vars
            md.shared = dict(first=v["sh_first"], count=v["sh_count"], prefix=v["sh_prefix"], members=v["sh_members"], start=v["sh_start"][:bs],
                             extra=v["sh_extra"][:bs], items=v["sh_items"], key_chunk_dev=v["sh_key_chunk"], key_chunk=1024,
                             max_members=0, max_prefix=0)
            self.reset_graph_shared()  # the capture's warm-up runs see a batch in which nothing is shared
            self.decode_graph_metadata_shared[bs] = md
        else:
            self.decode_graph_metadata[bs] = md
        self.forward_metadata = md

    def graph_can_share_prefix(self, bs: int) -> bool:
        return self.share_prefix_in_graphs and bs >= 2 and splits_for_batch(bs) >= 2

    def _graph_md(self, bs: int) -> HipAttnMetadata:
        return (self.decode_graph_metadata_shared if self._graph_use_shared else self.decode_graph_metadata)[bs]

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                                                seq_lens_cpu: torch.Tensor, forward_mode):
        """The kernels read req_pool_indices / seq_lens from the runner's static buffers; only the split size
        depends on the batch, and it lives in one device word (compare fa3_attn_backend.py:203-233, which
        copies a [bs, max_ctx] table every step)."""
        md = self._graph_md(bs)
        if forward_mode.is_decode():
            own_splits = md.max_splits
            if md.shared is not None:
                most_extra, start = self._refresh_graph_shared(bs, md)
                own_splits -= most_extra
                if most_extra:  # the gather kernel reads what lies behind the shared prefixes
                    n = seq_lens_cpu.numel()  # the unpadded batch
                    seq_lens_cpu = (seq_lens_cpu - torch.tensor(start[:n], dtype=seq_lens_cpu.dtype)).clamp_(min=1)
            # padded rows have length 1; they only lower the average, never the maximum
            chunk, ragged = self._decode_split(seq_lens_cpu, bs, own_splits)
            self.decode_graph_vars["chunk"].fill_(chunk)
            self.decode_graph_vars["plan_sorted"].fill_(1 if ragged else 0)
            self.refresh_decode_plan(bs, seq_lens)
        self.forward_metadata = md

    def refresh_decode_plan(self, bs: int, seq_lens: torch.Tensor) -> None:
        """Rewrite the static launch order from the lengths now in `seq_lens` (device).  The replay hook calls it;
        a runner whose captured step changes the lengths itself (device-side batch preparation) calls it once more
        inside the step, after the lengths are final."""
        md = self._graph_md(bs)
        if md.decode_plan is not None:
            ops.decode_plan(seq_lens[:bs], md.max_splits, chunk_dev=md.chunk_dev, out=md.decode_plan,
                            sorted_dev=self.decode_graph_vars["plan_sorted"],
                            start_lens=md.shared["start"] if md.shared is not None else None)

    def set_graph_shared_groups(self, groups, bs: int) -> bool:
        """Optional, before a replay: the batch's groups of requests on a shared radix prefix, [(member indices, prefix
        length)] (BatchInfo.shared_groups()), and the batch size of the graph it will run in.  Returns whether the
        replay should use the graph that was captured with the shared-prefix kernel (capture_shared_prefix) — the
        replay hook then refreshes that graph's static state.  Like set_graph_reuse_lens, not part of the
        reference's hook set; a runner that never calls it gets per-request prefix reads."""
        self._graph_groups = list(groups or [])
        self._graph_use_shared = bool(self._graph_groups) and self.share_prefix_reads and self.graph_can_share_prefix(bs)
        return self._graph_use_shared

    def reset_graph_shared(self) -> None:
        """Static group buffers back to "nothing is shared"."""
        v = self.decode_graph_vars
        for name in ("sh_first", "sh_count", "sh_prefix", "sh_members", "sh_start", "sh_extra"):
            v[name].zero_()
        v["sh_items"].fill_(-1)
        self._graph_shared_key = None

    def _refresh_graph_shared(self, bs: int, md: HipAttnMetadata):
        """Bring the static group buffers in line with the groups announced for this replay.  Returns (largest number
        of prefix slices of any group, per-request start positions).  Groups are dropped as a whole when their work
        list or their slices do not fit what was captured."""
        v = self.decode_graph_vars
        groups = [g for g in self._graph_groups if max(g[0]) < bs] if self.share_prefix_reads else []
        heads = self.num_heads // self.num_kv_heads
        key_chunk, items = 1024, None
        while groups:
            items = ops.shared_prefix_items(groups, heads, key_chunk, GRAPH_SHARED_ITEMS)
            if items is not None and max(-(-g[1] // key_chunk) for g in groups) <= md.max_splits - 1:
                break
            key_chunk *= 2
            if key_chunk > 65536:
                groups, items = [], None
        start, extra = [0] * bs, [0] * bs
        for members, plen in groups:
            for i in members:
                start[i], extra[i] = plen, -(-plen // key_chunk)
        key = (bs, tuple(groups), key_chunk)
        if key != self._graph_shared_key:
            i32 = dict(dtype=torch.int32)
            pad = lambda vals, n: torch.tensor(list(vals) + [0] * (n - len(vals)), **i32)
            n = v["sh_first"].numel()
            first, flat = _flatten_groups(groups)
            v["sh_first"].copy_(pad(first, n))
            v["sh_count"].copy_(pad([len(g[0]) for g in groups], n))
            v["sh_prefix"].copy_(pad([g[1] for g in groups], n))
            v["sh_members"].copy_(pad(flat, n))
            v["sh_start"].copy_(pad(start, v["sh_start"].numel()))
            v["sh_extra"].copy_(pad(extra, v["sh_extra"].numel()))
            v["sh_items"].copy_(torch.tensor(items if groups else [-1] * GRAPH_SHARED_ITEMS, **i32))
            v["sh_key_chunk"].fill_(key_chunk)
            self._graph_shared_key = key
        return max(extra) if groups else 0, start

    def set_graph_reuse_lens(self, shared_prefix_lens: Optional[torch.Tensor]) -> None:
        """Optional, before a replay: per-request shared-prefix lengths for the cache-policy hint (None = none).
        Not part of the reference's hook set; a runner that never calls it gets pure streaming reads."""
        buf = self.decode_graph_vars["reuse_lens"]
        buf.zero_()
        if shared_prefix_lens is not None:
            buf[: shared_prefix_lens.numel()].copy_(shared_prefix_lens)


def _flatten_groups(groups):
    """[(member indices, prefix length)] -> (offset of every group in the flat member list, the flat list)."""
    first, flat = [], []
    for members, _ in groups:
        first.append(len(flat))
        flat.extend(members)
    return first, flat


def _as_heads(q: torch.Tensor, num_heads: int, head_dim: int) -> torch.Tensor:
    """[T, H*D] (possibly a row-strided view) -> [T, H, D] without copying."""
    return q.view(q.shape[0], num_heads, head_dim) if q.dim() == 2 else q
