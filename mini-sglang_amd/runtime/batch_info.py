"""Batch tensors: the producer side of the hot path's inputs.

Same observable contract as the reference's BatchInfo (mini_sglang/managers/batch_info.py:24-302): the
fields a model forward reads (``forward_mode, input_ids, positions, seq_lens, seq_lens_cpu, input_seq_lens,
prefix_seq_lens, out_cache_loc, req_pool_indices``) have the same dtypes, shapes and VALUES for the same
request sequence — pages are taken from the allocator in the same order, so ``req_to_token`` comes out
identical — but the table is written with one scatter per step from host-built index arrays instead of
per-request device slices, and decode never reads the table back (the reference reads
``req_to_token[idx, len-1] + 1`` per request, batch_info.py:286-291; here each request remembers its last
slot on the host).
"""
from __future__ import annotations

import dataclasses
from enum import IntEnum, auto
from itertools import count
from typing import List, Optional

import torch


class ForwardMode(IntEnum):
    EXTEND = auto()
    DECODE = auto()

    def is_extend(self) -> bool:
        return self == ForwardMode.EXTEND

    def is_decode(self) -> bool:
        return self == ForwardMode.DECODE


@dataclasses.dataclass
class SamplingParams:
    """mini_sglang/managers/sampling_params.py:4-10."""
    temperature: float = 0.0
    max_new_tokens: int = 64
    ignore_eos: bool = False
    eos_token_id: int = -1


class Req:
    """The per-request state the batch preparation needs (mini_sglang/managers/req_info.py:64-100)."""
    _counter = count()

    def __init__(self, rid, token_ids: List[int], sampling_params: Optional[SamplingParams] = None, req_pool_idx: int = -1):
        self.rid = rid
        self.req_id = next(Req._counter)
        self.token_ids = list(token_ids)
        self.num_prompt_tokens = len(token_ids)
        self.sampling_params = sampling_params or SamplingParams()
        self.req_pool_idx = req_pool_idx
        # slots of an already-cached prefix (set by a prefix cache before prepare_for_extend); int32, CPU or device
        self.prefix_indices: torch.Tensor = torch.empty(0, dtype=torch.int32)
        self.last_slot: int = -1  # pool slot of the request's last stored token
        self.last_node = None     # radix-cache node that ends the cached prefix (a handle; None = no cache)
        # identity of the cached prefix the request was prefilled on: (length, first slot, last slot); a slot holds
        # one token of one radix path, so equal keys mean the same slots at positions [0, length)
        self.shared_key = None

    @property
    def last_token_id(self) -> int:
        return self.token_ids[-1]

    @property
    def num_tokens(self) -> int:
        return len(self.token_ids)

    def __len__(self) -> int:
        return len(self.token_ids)


@dataclasses.dataclass
class BatchInfo:
    forward_mode: ForwardMode = None
    input_ids: torch.Tensor = None          # int32 [sum(input_seq_len)]
    positions: torch.Tensor = None          # int32 [sum(input_seq_len)]
    input_seq_lens: torch.Tensor = None     # int32 [bs] (extend only)
    input_seq_lens_max: int = 0
    seq_lens: torch.Tensor = None           # int32 [bs] device
    seq_lens_cpu: torch.Tensor = None       # int32 [bs] host
    prefix_seq_lens: torch.Tensor = None    # int32 [bs] (extend only)
    out_cache_loc: torch.Tensor = None      # int32 [num new tokens]
    req_pool_indices: torch.Tensor = None   # int32 [bs]
    # int32 [bs]: leading positions whose slots came from the prefix cache, i.e. may be shared with other requests
    # (a cache-policy hint for decode attention; None = unknown / nothing shared)
    shared_prefix_lens: torch.Tensor = None
    req_to_token_pool: object = None
    page_allocator: object = None
    kv_cache_pool: object = None
    tree_cache: object = None
    attn_backend: object = None
    reqs: List[Req] = None
    device: str = "cuda"

    @classmethod
    def init_new(cls, reqs, req_to_token_pool, page_allocator, kv_cache_pool, tree_cache=None):
        return cls(reqs=reqs, req_to_token_pool=req_to_token_pool, page_allocator=page_allocator,
                   kv_cache_pool=kv_cache_pool, tree_cache=tree_cache, device=req_to_token_pool.device)

    @property
    def batch_size(self) -> int:
        return len(self.reqs)

    def alloc_page_slots(self, num_pages: int) -> torch.Tensor:
        need = num_pages * self.page_allocator.page_size
        if self.tree_cache is not None and self.page_allocator.available_size() < need:
            self.tree_cache.evict(need - self.page_allocator.available_size())
        return self.page_allocator.alloc(num_pages)

    # ------------------------------------------------------------------ extend
    def prepare_for_extend(self) -> None:
        """batch_info.py:135-245.  Page order: ps == 1 -> one page per new token in batch order; ps > 1 ->
        ceil(new / ps) pages per request in batch order, tokens laid out contiguously inside pages; prefixes
        must be page aligned (batch_info.py:208-210)."""
        self.forward_mode = ForwardMode.EXTEND
        reqs = self.reqs
        bs = len(reqs)
        ps = self.page_allocator.page_size
        pool = self.req_to_token_pool
        rows = pool.alloc(bs)
        for r, row in zip(reqs, rows):
            r.req_pool_idx = row
        seq_lens = [len(r.token_ids) for r in reqs]
        prefix_lens = [int(r.prefix_indices.numel()) for r in reqs]
        extend_lens = [s - p for s, p in zip(seq_lens, prefix_lens)]
        total_new = sum(extend_lens)
        self.input_seq_lens_max = max(extend_lens)

        if ps == 1:
            new_slots = self.alloc_page_slots(total_new).to(torch.int32)
        else:
            for p in prefix_lens:
                if p % ps != 0:
                    raise AssertionError(f"Prefix length {p} is not page-aligned")
            pages_per_req = [-(-n // ps) for n in extend_lens]
            pages = self.alloc_page_slots(sum(pages_per_req)).to(torch.int64)
            expanded = (pages[:, None] * ps + torch.arange(ps, dtype=torch.int64)[None, :])  # [pages, ps]
            keep = torch.zeros(expanded.shape, dtype=torch.bool)
            pg = 0
            for n, npg in zip(extend_lens, pages_per_req):
                flat = keep[pg: pg + npg].view(-1)
                flat[:n] = True
                pg += npg
            new_slots = expanded[keep].to(torch.int32)

        # host-built scatter: (row, column) -> slot for every token of every request (prefix + new)
        row_idx, col_idx, vals = [], [], []
        pos = 0
        for i, r in enumerate(reqs):
            P, n = prefix_lens[i], extend_lens[i]
            if P:
                r.shared_key = (P, int(r.prefix_indices[0]), int(r.prefix_indices[P - 1]))
                row_idx.append(torch.full((P,), rows[i], dtype=torch.int64))
                col_idx.append(torch.arange(P, dtype=torch.int64))
                vals.append(r.prefix_indices.to("cpu", torch.int32))
            row_idx.append(torch.full((n,), rows[i], dtype=torch.int64))
            col_idx.append(torch.arange(P, P + n, dtype=torch.int64))
            vals.append(new_slots[pos: pos + n])
            r.last_slot = int(new_slots[pos + n - 1])
            pos += n
        dev = self.device
        pool.scatter_tokens(torch.cat(row_idx), torch.cat(col_idx), torch.cat(vals))

        self.input_ids = torch.tensor([t for r, p in zip(reqs, prefix_lens) for t in r.token_ids[p:]], dtype=torch.int32).to(dev)
        self.positions = torch.cat([torch.arange(p, s, dtype=torch.int32) for p, s in zip(prefix_lens, seq_lens)]).to(dev)
        self.req_pool_indices = torch.tensor(rows, dtype=torch.int32).to(dev)
        self.seq_lens_cpu = torch.tensor(seq_lens, dtype=torch.int32)
        self.seq_lens = self.seq_lens_cpu.to(dev)
        self.prefix_seq_lens = torch.tensor(prefix_lens, dtype=torch.int32).to(dev)
        self.input_seq_lens = self.seq_lens - self.prefix_seq_lens
        self.shared_prefix_lens = self.prefix_seq_lens.clone() if any(prefix_lens) else None
        self.out_cache_loc = new_slots.to(dev)

    # ------------------------------------------------------------------ decode
    def prepare_for_decode(self) -> None:
        """batch_info.py:247-302: one new token per request at position old_len; ps > 1 takes a fresh page when
        old_len % ps == 0, else the slot after the request's last one."""
        self.forward_mode = ForwardMode.DECODE
        reqs = self.reqs
        bs = len(reqs)
        ps = self.page_allocator.page_size
        dev = self.device
        old_lens = self.seq_lens_cpu.clone()
        self.seq_lens_cpu = self.seq_lens_cpu + 1
        self.seq_lens = self.seq_lens + 1  # new tensors: captured graphs hold their own static copies
        self.prefix_seq_lens = None
        self.input_seq_lens = None

        if ps == 1:
            new_slots = self.alloc_page_slots(bs).to(torch.int32)
        else:
            old = old_lens.tolist()
            need_page = [l % ps == 0 for l in old]
            pages = self.alloc_page_slots(sum(need_page)).tolist()
            it = iter(pages)
            new_slots = torch.tensor([next(it) * ps if np_ else r.last_slot + 1 for r, np_ in zip(reqs, need_page)],
                                     dtype=torch.int32)
        for r, s in zip(reqs, new_slots.tolist()):
            r.last_slot = s
        self.out_cache_loc = new_slots.to(dev)
        self.positions = old_lens.to(dev)
        self.req_to_token_pool.scatter_tokens(torch.tensor([r.req_pool_idx for r in reqs], dtype=torch.int64), old_lens.long(), new_slots)
        self.input_ids = torch.tensor([r.last_token_id for r in reqs], dtype=torch.int32).to(dev)

    # ------------------------------------------------------------------ bookkeeping used by callers
    def shared_groups(self) -> List[tuple]:
        """Requests of the batch whose page-table rows start with the same cached prefix, as (batch indices of the
        members, prefix length) — what the attention backend may compute once per group.  Members need not be
        neighbours; groups come in the order of their first member."""
        by_key = {}
        for i, r in enumerate(self.reqs or []):
            if r.shared_key is not None:
                by_key.setdefault(r.shared_key, []).append(i)
        return [(tuple(m), key[0]) for key, m in by_key.items() if len(m) >= 2]

    def merge_batch(self, other: "BatchInfo") -> None:
        """batch_info.py:304-317."""
        self.reqs.extend(other.reqs)
        self.input_ids = torch.cat([self.input_ids, other.input_ids])
        self.positions = torch.cat([self.positions, other.positions])
        self.seq_lens = torch.cat([self.seq_lens, other.seq_lens])
        self.seq_lens_cpu = torch.cat([self.seq_lens_cpu, other.seq_lens_cpu])
        self.req_pool_indices = torch.cat([self.req_pool_indices, other.req_pool_indices])
        self.out_cache_loc = torch.cat([self.out_cache_loc, other.out_cache_loc])
        if self.shared_prefix_lens is not None or other.shared_prefix_lens is not None:
            a = self.shared_prefix_lens if self.shared_prefix_lens is not None else torch.zeros(
                len(self.reqs) - len(other.reqs), dtype=torch.int32, device=self.seq_lens.device)
            b = other.shared_prefix_lens if other.shared_prefix_lens is not None else torch.zeros(
                len(other.reqs), dtype=torch.int32, device=self.seq_lens.device)
            self.shared_prefix_lens = torch.cat([a, b])

    def filter_reqs(self, keep_indices: List[int]) -> None:
        """batch_info.py:319-342 (keep form)."""
        self.reqs = [self.reqs[i] for i in keep_indices]
        idx = torch.tensor(keep_indices, dtype=torch.int64)
        self.seq_lens = self.seq_lens[idx.to(self.seq_lens.device)]
        self.seq_lens_cpu = self.seq_lens_cpu[idx]
        self.req_pool_indices = self.req_pool_indices[idx.to(self.req_pool_indices.device)]
        if self.shared_prefix_lens is not None:
            self.shared_prefix_lens = self.shared_prefix_lens[idx.to(self.shared_prefix_lens.device)]
