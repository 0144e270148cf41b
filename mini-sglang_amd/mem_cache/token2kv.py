"""Slot pool for K/V and the page allocator.

Tensor contract of the reference's MHAKVPool (mini_sglang/mem_cache/token2kv.py:112-151): per layer a
K and a V buffer indexed ``[slot, kv_head, head_dim]`` with ``size + page_size`` slots, slots
``0 .. page_size-1`` being a dummy page that padded rows may write (the allocator never hands out
page 0, token2kv.py:209-212).

Layout on MI355X: one allocation per layer of shape ``[slots, 2, Hkv, D]``; ``k_buffer[l]`` and
``v_buffer[l]`` are the strided views ``[:, 0]`` and ``[:, 1]``.  A token's K and V rows are then
adjacent (4 KiB at 8 kv heads) and a page is one contiguous run, which is what the decode kernel's
gather wants from HBM.  ``layout="split"`` gives the reference's two separate dense buffers.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from .. import ops


class MHAKVPool:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int, layer_num: int,
                 device: str, layout: str = "interleaved"):
        if layout not in ("interleaved", "split"):
            raise ValueError(f"unknown KV layout {layout!r}")
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.head_num = head_num
        self.head_dim = head_dim
        self.layer_num = layer_num
        self.device = device
        self.layout = layout
        slots = size + page_size
        self.k_buffer: List[torch.Tensor] = []
        self.v_buffer: List[torch.Tensor] = []
        for _ in range(layer_num):
            if layout == "interleaved":
                kv = torch.zeros((slots, 2, head_num, head_dim), dtype=dtype, device=device)
                self.k_buffer.append(kv[:, 0])
                self.v_buffer.append(kv[:, 1])
            else:
                self.k_buffer.append(torch.zeros((slots, head_num, head_dim), dtype=dtype, device=device))
                self.v_buffer.append(torch.zeros((slots, head_num, head_dim), dtype=dtype, device=device))
        self.mem_usage = 2 * layer_num * slots * head_num * head_dim * torch.empty((), dtype=dtype).element_size() / (1 << 30)

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self.k_buffer[layer_id]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self.v_buffer[layer_id]

    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.k_buffer[layer_id], self.v_buffer[layer_id]

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
        """k_buffer[layer][loc] = cache_k, v likewise (token2kv.py:133-142), as one HIP scatter kernel."""
        lid = layer.layer_id
        ops.store_kv(cache_k.view(-1, self.head_num, self.head_dim), cache_v.view(-1, self.head_num, self.head_dim), loc,
                     self.k_buffer[lid], self.v_buffer[lid])


class PageAllocator:
    """Free list of page ids 1..page_num; page p owns slots [p*page_size, (p+1)*page_size)
    (mini_sglang/mem_cache/token2kv.py:154-212).  Kept on the host: allocation is control-plane work.  The list itself
    lives in libmsgl_host.so (a ring buffer behind ``msgl_pages_*``): alloc takes from the front, free appends at the
    back, as the reference's tensor slicing / concatenation does, at the cost of the pages moved instead of the length
    of the list.  ``free_page`` stays readable and assignable as a tensor, which is how the reference's callers (and
    this package's synthetic workloads) inspect and lay out a pool."""

    def __init__(self, page_num: int, page_size: int, device: str = "cpu"):
        from .. import _capi_host

        self._lib = _capi_host.lib
        self.page_num = page_num
        self.page_size = page_size
        self.device = device
        self._h = self._lib.msgl_pages_create(int(page_num), int(page_size))
        if not self._h:
            raise ValueError(f"PageAllocator: page_num={page_num} page_size={page_size}")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.msgl_pages_destroy(h)

    @staticmethod
    def _ptr(t: torch.Tensor):
        import ctypes

        return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int32))

    def free_count(self) -> int:
        return int(self._lib.msgl_pages_free_count(self._h))

    def available_size(self) -> int:
        return self.free_count() * self.page_size

    def alloc(self, num: int) -> torch.Tensor:
        out = torch.empty(max(int(num), 0), dtype=torch.int32)
        got = self._lib.msgl_pages_alloc(self._h, int(num), self._ptr(out))
        if got != num:
            raise RuntimeError(
                f"Not enough free pages available. need {num}, free {self.free_count()}, total {self.page_num}")
        return out

    def free(self, page_ids: torch.Tensor) -> None:
        ids = page_ids.to("cpu", torch.int32).contiguous().view(-1)
        if self._lib.msgl_pages_free(self._h, self._ptr(ids), ids.numel()) < 0:
            raise ValueError(f"PageAllocator.free: page id outside 1..{self.page_num}")

    def clear(self) -> None:
        self._lib.msgl_pages_clear(self._h)

    @property
    def free_page(self) -> torch.Tensor:
        """The free list in allocation order (a copy)."""
        n = self.free_count()
        out = torch.empty(n, dtype=torch.int32)
        self._lib.msgl_pages_peek(self._h, self._ptr(out), n)
        return out

    @free_page.setter
    def free_page(self, pages: torch.Tensor) -> None:
        ids = pages.to("cpu", torch.int32).contiguous().view(-1)
        if self._lib.msgl_pages_assign(self._h, self._ptr(ids), ids.numel()) < 0:
            raise ValueError(f"PageAllocator.free_page: page id outside 1..{self.page_num}")
