"""Request -> slot table.  Same tensor contract as the reference's ReqToTokenPool
(mini_sglang/mem_cache/req2token.py:28-75): ``req_to_token[r, t]`` (int32) is the pool slot of token t
of the request holding row r; rows of requests that share a cached prefix hold the same slots.

The kernels read ONLY this table.  The reference additionally keeps ``req_to_page``, but its writer
strides by pages instead of tokens (req2token.py:86-91) and leaves wrong entries for multi-page
prefixes, so page ids are derived here from the token table on demand instead of being stored.
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

    def page_table(self, req_indices: torch.Tensor, num_pages: int) -> torch.Tensor:
        """[len(req_indices), num_pages] page ids, derived from the first slot of every page."""
        ps = self.page_size
        return self.req_to_token[req_indices.long(), : num_pages * ps : ps] // ps
