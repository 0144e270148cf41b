"""Request -> slot table.  Same tensor contract as the reference's ReqToTokenPool
(mini_sglang/mem_cache/req2token.py:28-91): ``req_to_token[r, t]`` (int32) is the pool slot of token t
of the request holding row r; rows of requests that share a cached prefix hold the same slots.
``req_to_page[r, i]`` is the page of tokens i * page_size .. of that request (the same tensor as
``req_to_token`` when page_size == 1, req2token.py:43-51).

The kernels read only the token table.  The page table is kept for callers that want one entry per page
(the reference's FA3 / FA2 backends do): every writer of the token table in this package writes it too —
``write_tokens`` here, the batch producers' scatters, and the device-side step preparation.  The reference's own
writer takes every page_size-th entry of an index vector that already counts pages (req2token.py:86-91), which
leaves wrong ids from the second page of a prefix on; here page i of a write is ``values[i * page_size] // page_size``.
"""
from __future__ import annotations

from typing import List, Union

import torch


class ReqToTokenPool:
    def __init__(self, size: int, max_tokens: int, page_size: int, device: str):
        if max_tokens % page_size != 0:
            raise ValueError("max_tokens must be a multiple of page_size")
        self.size = size
        self.max_tokens = max_tokens
        self.page_size = page_size
        self.max_page_num = max_tokens // page_size
        self.device = device
        self.req_to_token = torch.zeros((size, max_tokens), dtype=torch.int32, device=device)
        self.req_to_page = (self.req_to_token if page_size == 1
                            else torch.zeros((size, self.max_page_num), dtype=torch.int32, device=device))
        self.free_slots: List[int] = list(range(size))

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, num: int) -> List[int]:
        if num > len(self.free_slots):
            raise RuntimeError("Not enough free slots available.")
        got, self.free_slots = self.free_slots[:num], self.free_slots[num:]
        return got

    def free(self, indices: Union[int, List[int]]) -> None:
        if isinstance(indices, int):
            indices = [indices]
        self.free_slots.extend(indices)

    def clear(self) -> None:
        self.free_slots = list(range(self.size))

    def write_tokens(self, req_idx: int, indices: slice, values: torch.Tensor) -> None:
        """Slots of the token positions ``indices`` of one request (req2token.py:77-91), page table included."""
        self.req_to_token[req_idx, indices] = values
        ps = self.page_size
        if ps > 1:
            if indices.start % ps != 0 or indices.stop % ps != 0:
                raise AssertionError("Token indices must be page-aligned when page_size > 1.")
            self.req_to_page[req_idx, indices.start // ps: indices.stop // ps] = values[::ps] // ps

    def scatter_tokens(self, rows: torch.Tensor, cols: torch.Tensor, slots: torch.Tensor) -> None:
        """req_to_token[rows[j], cols[j]] = slots[j] for all j in one scatter, and the page table entries of the
        positions among them that start a page.  HOST tensors (int64 indices, int32 slots): the subset that starts a
        page is picked here, where that costs no device synchronisation, and both scatters then run on the device."""
        dev = self.device
        self.req_to_token[(rows.to(dev), cols.to(dev))] = slots.to(dev)
        ps = self.page_size
        if ps > 1:
            first = (cols % ps) == 0
            if bool(first.any()):
                self.req_to_page[(rows[first].to(dev), (cols[first] // ps).to(dev))] = (slots[first] // ps).to(dev)

    def load_rows(self, rows: torch.Tensor, table: torch.Tensor) -> None:
        """Whole rows of the token table at once (synthetic workloads); the page table follows."""
        self.req_to_token[rows] = table
        if self.page_size > 1:
            self.req_to_page[rows] = table[:, :: self.page_size] // self.page_size

    def page_table(self, req_indices: torch.Tensor, num_pages: int) -> torch.Tensor:
        """[len(req_indices), num_pages] page ids, derived from the first slot of every page."""
        ps = self.page_size
        return self.req_to_token[req_indices.long(), : num_pages * ps : ps] // ps
