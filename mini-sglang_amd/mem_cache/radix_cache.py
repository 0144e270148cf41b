"""Radix prefix cache with the reference's interface (mini_sglang/mem_cache/radix_cache.py:53-418,
base_cache.py:9-48) on top of the C++ tree in libmsgl_host.so.

Differences from the reference, all on the host side of the path:
- the tree lives in C++ and keeps KV slot indices in host memory; ``match_prefix`` copies the matched run to the
  pool's device once, instead of concatenating per-node device tensors;
- ``last_node`` is an integer handle (0 = root), not a Python object;
- ``evict`` and the request-level calls return PAGES to the allocator for every page size (slot // page_size of
  each page's first slot, read from ``req_to_token``); the reference frees slot lists (radix_cache.py:383) and
  reads ``req_to_page`` (radix_cache.py:291-296), both wrong for page_size > 1.
"""
from __future__ import annotations

import ctypes
from typing import List, NamedTuple, Optional

import numpy as np
import torch

from .. import _capi_host

_lib = _capi_host.lib
_I32P, _I64P = _capi_host._I32P, _capi_host._I64P


class MatchResult(NamedTuple):
    match_indices: torch.Tensor   # int32 slots of the matched prefix, on the pool's device
    last_node: int                # handle for inc_lock_ref / dec_lock_ref


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p32(a: np.ndarray):
    return a.ctypes.data_as(_I32P)


class RadixCache:
    def __init__(self, req_to_token_pool, kv_cache_pool, page_allocator, page_size: int):
        self.req_to_token_pool = req_to_token_pool
        self.kv_cache_pool = kv_cache_pool
        self.page_allocator = page_allocator
        self.page_size = page_size
        self.device = getattr(kv_cache_pool, "device", None) or getattr(req_to_token_pool, "device", "cpu")
        self._h = _lib.msgl_radix_create(page_size)
        if not self._h:
            raise ValueError(f"page_size={page_size}")
        self.root = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.msgl_radix_destroy(h)

    def reset(self) -> None:
        _lib.msgl_radix_reset(self._h)

    # ------------------------------------------------------------------ tree
    def match_prefix(self, key: List[int], **kwargs) -> MatchResult:
        k = _i32(key)
        out = np.empty(max(1, k.size), dtype=np.int32)
        node = ctypes.c_int64(0)
        n = _lib.msgl_radix_match_prefix(self._h, _p32(k), k.size, _p32(out), out.size, ctypes.byref(node))
        assert n >= 0
        return MatchResult(torch.from_numpy(out[:n].copy()).to(self.device), int(node.value))

    def insert(self, key: List[int], value=None) -> int:
        k = _i32(key)
        if value is None:
            v = k
        else:
            v = _i32(value.cpu().numpy() if isinstance(value, torch.Tensor) else value)
        assert v.size == k.size, "one slot per token"
        return int(_lib.msgl_radix_insert(self._h, _p32(k), _p32(v), k.size))

    def inc_lock_ref(self, node: Optional[int]) -> None:
        if node is not None and _lib.msgl_radix_inc_lock_ref(self._h, node) != 0:
            raise KeyError(f"unknown radix node {node}")

    def dec_lock_ref(self, node: Optional[int]) -> None:
        if node is not None and _lib.msgl_radix_dec_lock_ref(self._h, node) != 0:
            raise KeyError(f"unknown radix node {node}")

    def evict_nodes(self, num_tokens: int) -> List[np.ndarray]:
        """Evict until at least num_tokens slots are gone; the evicted nodes' slot runs in eviction order."""
        cap, ncap = max(1, self.evictable_size()), max(1, int(_lib.msgl_radix_node_count(self._h)))
        slots, lens = np.empty(cap, dtype=np.int32), np.empty(ncap, dtype=np.int64)
        n_nodes = ctypes.c_int64(0)
        done = _lib.msgl_radix_evict(self._h, int(num_tokens), _p32(slots), cap, lens.ctypes.data_as(_I64P), ncap, ctypes.byref(n_nodes))
        assert done >= 0
        out, pos = [], 0
        for i in range(n_nodes.value):
            out.append(slots[pos: pos + lens[i]].copy())
            pos += int(lens[i])
        return out

    def evict(self, num_tokens: int) -> None:
        ps = self.page_size
        for run in self.evict_nodes(num_tokens):
            if self.page_allocator is not None:
                self.page_allocator.free(torch.from_numpy(run[::ps] // ps))

    def evictable_size(self) -> int:
        return int(_lib.msgl_radix_evictable_size(self._h))

    def protected_size(self) -> int:
        return int(_lib.msgl_radix_protected_size(self._h))

    # ------------------------------------------------------------------ request level
    def _cache_req(self, req) -> int:
        """Shared first half of cache_unfinished_req / cache_finished_req (radix_cache.py:272-296, 336-360): insert
        the page-aligned tokens with their slots and give back the pages this request holds for tokens that were
        already in the tree.  Returns the page-aligned length of the request's old prefix."""
        ps = self.page_size
        tokens_len = len(req.token_ids) - 1          # the last token has no KV yet
        aligned = tokens_len // ps * ps
        row = self.req_to_token_pool.req_to_token[req.req_pool_idx, :aligned].to("cpu", torch.int32).numpy()
        new_prefix_len = self.insert(req.token_ids[:aligned], row)
        old_aligned = int(req.prefix_indices.numel()) // ps * ps
        dup = row[old_aligned:new_prefix_len:ps] // ps
        if dup.size:
            self.page_allocator.free(torch.from_numpy(dup.copy()))
        return old_aligned

    def cache_unfinished_req(self, req) -> None:
        """radix_cache.py:270-332: after a prefill (chunk), move the request's lock to the node that now ends its
        cached tokens and point its page-table row at the tree's slots."""
        ps = self.page_size
        tokens_len = len(req.token_ids) - 1
        aligned = tokens_len // ps * ps
        old_aligned = self._cache_req(req)
        new_indices, new_last = self.match_prefix(req.token_ids[:aligned])
        r2t = self.req_to_token_pool.req_to_token
        if new_indices.numel() > old_aligned:
            r2t[req.req_pool_idx, old_aligned: new_indices.numel()] = new_indices[old_aligned:].to(r2t.device)
        self.dec_lock_ref(getattr(req, "last_node", None))
        self.inc_lock_ref(new_last)
        if ps == 1:
            req.prefix_indices = new_indices
        else:
            tail = r2t[req.req_pool_idx, new_indices.numel(): tokens_len].to(new_indices.device, torch.int32)
            req.prefix_indices = torch.cat([new_indices, tail])
        req.last_node = new_last

    def cache_finished_req(self, req) -> None:
        """radix_cache.py:334-365: keep the finished request's tokens in the tree, release what the tree does not
        keep (duplicates, the incomplete last page), its table row and its lock."""
        ps = self.page_size
        tokens_len = len(req.token_ids) - 1
        self._cache_req(req)
        if ps > 1 and tokens_len % ps != 0:
            first = int(self.req_to_token_pool.req_to_token[req.req_pool_idx, tokens_len // ps * ps])
            self.page_allocator.free(torch.tensor([first // ps], dtype=torch.int32))
        self.req_to_token_pool.free(req.req_pool_idx)
        self.dec_lock_ref(getattr(req, "last_node", None))
