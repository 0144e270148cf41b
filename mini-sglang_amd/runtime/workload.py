"""Synthetic page tables and pools shaped like what the scheduler + radix cache hand to the hot path.

The reference builds ``req_to_token`` in BatchInfo.prepare_for_extend / prepare_for_decode
(mini_sglang/managers/batch_info.py:135-302) from pages handed out by PageAllocator and from prefix
slots matched in the radix cache (mini_sglang/mem_cache/radix_cache.py:115-168), which makes rows of
different requests share the slots of a common, page-aligned prefix.  This module produces tables
with exactly those properties (page-granular allocation from ids 1.., contiguous slots inside a
page, shared page-aligned prefixes, fully scattered page order) without running a scheduler.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import synth


@dataclass
class PageTable:
    req_to_token: torch.Tensor  # int32 [bs, max_ctx] (CPU)
    seq_lens: torch.Tensor      # int32 [bs] (CPU)
    pages_used: int             # distinct pages referenced
    unique_tokens: int          # distinct slots referenced (shared prefixes counted once)
    page_size: int


def make_page_table(seq_lens: Sequence[int], page_size: int, pool_pages: int, seed: int, *,
                    groups: Optional[Sequence[int]] = None, group_prefix_lens: Optional[Sequence[int]] = None,
                    scatter: bool = True, max_ctx: Optional[int] = None) -> PageTable:
    """Allocate pages for ``len(seq_lens)`` requests out of page ids 1..pool_pages.

    groups[r] / group_prefix_lens[g]: request r shares the first group_prefix_lens[g] tokens (a multiple of
    page_size, at most its own length) with every other request of group g — the rows hold the SAME slots
    for that prefix, as after a radix-cache hit.  scatter=True draws pages in a seeded random order (worst
    case for DRAM locality); False hands them out in ascending order like a fresh allocator.
    """
    bs = len(seq_lens)
    ps = page_size
    need_ctx = max(seq_lens) if bs else 0
    max_ctx = max_ctx or -(-need_ctx // ps) * ps
    order = (synth.permutation(pool_pages, seed) + 1) if scatter else torch.arange(1, pool_pages + 1)
    order = order.to(torch.int64)
    cursor = 0
    table = torch.zeros((bs, max_ctx), dtype=torch.int32)
    off = torch.arange(ps, dtype=torch.int64)
    prefix_pages = {}
    unique = 0

    def take(n: int) -> torch.Tensor:
        nonlocal cursor
        if cursor + n > pool_pages:
            raise RuntimeError(f"page pool exhausted: need {cursor + n} of {pool_pages} pages")
        got = order[cursor: cursor + n]
        cursor += n
        return got

    for r, S in enumerate(seq_lens):
        npages = -(-S // ps)
        shared = torch.empty(0, dtype=torch.int64)
        if groups is not None:
            g = groups[r]
            plen = group_prefix_lens[g]
            if plen % ps != 0:
                raise ValueError("shared prefixes must be page aligned")
            if plen > S:
                raise ValueError("a request cannot be shorter than its group's prefix")
            if g not in prefix_pages:
                prefix_pages[g] = take(plen // ps)
                unique += plen
            shared = prefix_pages[g]
        own = take(npages - shared.numel())
        unique += S - shared.numel() * ps
        pages = torch.cat([shared, own])
        slots = (pages[:, None] * ps + off[None, :]).reshape(-1)[:S]
        table[r, :S] = slots.to(torch.int32)
    return PageTable(table, torch.tensor(list(seq_lens), dtype=torch.int32), cursor, unique, ps)


def mixed_lengths(bs: int, lo: int, hi: int, seed: int) -> List[int]:
    """Log-uniform sequence lengths in [lo, hi]."""
    import math

    u = synth.uniform_int(bs, seed, 0, 1 << 20).to(torch.float64) / float(1 << 20)
    return [int(round(math.exp(math.log(lo) + float(x) * (math.log(hi) - math.log(lo))))) for x in u]
