"""CPU restatement of the reference's radix prefix cache (mini_sglang/mem_cache/radix_cache.py) — TEST
INFRASTRUCTURE: only tests/ may import it; the product is the C++ tree behind include/msgl_host.h.

Plain Python lists and ints.  Parity is pinned: tests/test_radix_cache.py replays the op sequences of
tests/golden/radix_tree.pt / radix_reqs.pt (outputs of the reference's own RadixCache, produced by
oracle/gen_golden_radix.py) and requires identical results, including WHICH nodes eviction picks and in what
order.  That order depends on the reference's access clock, so the clock ticks here at exactly the same
events: node creation (radix_cache.py:30-38) and every update_access_time() call in match_prefix (139-147)
and insert (186-191).

One deliberate difference, page_size > 1 only: evict() reports the evicted node's PAGES (slot // page_size of
every page_size-th slot); the reference hands the slot list itself to PageAllocator.free (radix_cache.py:383),
which corrupts the free list whenever page_size > 1.
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Optional, Tuple


class Node:
    __slots__ = ("children", "parent", "key", "value", "lock_ref", "time")

    def __init__(self, clock: "Clock"):
        self.children: Dict[Tuple[int, ...], "Node"] = {}
        self.parent: Optional["Node"] = None
        self.key: List[int] = []
        self.value: List[int] = []
        self.lock_ref = 0
        self.time = clock.tick()           # radix_cache.py:37

    def __lt__(self, other: "Node") -> bool:   # radix_cache.py:42-43 (heap order of evict)
        return self.time < other.time


class Clock:
    def __init__(self):
        self.t = 0

    def tick(self) -> int:
        self.t += 1
        return self.t


class RadixOracle:
    def __init__(self, page_size: int):
        self.ps = page_size
        self.clock = Clock()
        self.reset()

    def reset(self):                         # radix_cache.py:77-83
        self.root = Node(self.clock)
        self.root.lock_ref = 1
        self.evictable = 0
        self.protected = 0

    # ---- helpers ----
    def _child_key(self, key: List[int]) -> Tuple[int, ...]:       # radix_cache.py:85-86
        return tuple(key[: self.ps])

    def _match_len(self, a: List[int], b: List[int]) -> int:       # radix_cache.py:88-96, page by page
        n, i, ps = min(len(a), len(b)), 0, self.ps
        while i < n and a[i: i + ps] == b[i: i + ps]:
            i += ps
        return i

    def _split(self, child: Node, pos: int) -> Node:               # radix_cache.py:98-115
        new = Node(self.clock)
        new.children = {self._child_key(child.key[pos:]): child}
        new.parent = child.parent
        new.key, new.value = child.key[:pos], child.value[:pos]
        new.lock_ref = child.lock_ref
        child.parent = new
        child.key, child.value = child.key[pos:], child.value[pos:]
        new.parent.children[self._child_key(new.key)] = new
        return new

    # ---- tree API ----
    def match_prefix(self, key: List[int]) -> Tuple[List[int], Node]:    # radix_cache.py:117-168
        if len(key) == 0:
            return [], self.root
        if self.ps != 1:
            key = key[: len(key) // self.ps * self.ps]
        node = self.root
        node.time = self.clock.tick()
        out: List[int] = []
        ck = self._child_key(key)
        while len(key) > 0 and ck in node.children:
            child = node.children[ck]
            child.time = self.clock.tick()
            n = self._match_len(key, child.key)
            if n < len(child.key):
                node = self._split(child, n)
                out += node.value
                break
            out += child.value
            node = child
            key = key[n:]
            if len(key) > 0:
                ck = self._child_key(key)
        return out, node

    def insert(self, key: List[int], value: List[int]) -> int:           # radix_cache.py:170-214
        if len(key) == 0:
            return 0
        node = self.root
        node.time = self.clock.tick()
        ck = self._child_key(key)
        total = 0
        while len(key) > 0 and ck in node.children:
            node = node.children[ck]
            node.time = self.clock.tick()
            n = self._match_len(key, node.key)
            total += n
            key, value = key[n:], value[n:]
            if n < len(node.key):
                node = self._split(node, n)
            if len(key) > 0:
                ck = self._child_key(key)
        if len(key):
            new = Node(self.clock)
            new.parent, new.key, new.value = node, list(key), list(value)
            node.children[ck] = new
            self.evictable += len(value)
        assert total % self.ps == 0
        return total

    def inc_lock_ref(self, node: Optional[Node]):                         # radix_cache.py:216-226
        while node is not None and node is not self.root:
            if node.lock_ref == 0:
                self.evictable -= len(node.value)
                self.protected += len(node.value)
            node.lock_ref += 1
            node = node.parent

    def dec_lock_ref(self, node: Optional[Node]):                         # radix_cache.py:228-238
        while node is not None and node is not self.root:
            node.lock_ref -= 1
            if node.lock_ref == 0:
                self.evictable += len(node.value)
                self.protected -= len(node.value)
            node = node.parent

    def evict(self, num_tokens: int) -> List[List[int]]:                  # radix_cache.py:367-390
        """Returns the value lists of the evicted nodes in eviction order (what the reference frees)."""
        leaves, stack = [], [self.root]
        while stack:                                                      # radix_cache.py:398-410
            cur = stack.pop()
            if not cur.children:
                leaves.append(cur)
            else:
                stack.extend(cur.children.values())
        heapq.heapify(leaves)
        freed: List[List[int]] = []
        done = 0
        while done < num_tokens and leaves:
            x = heapq.heappop(leaves)
            if x is self.root:
                break
            if x.lock_ref > 0:
                continue
            freed.append(list(x.value))
            done += len(x.value)
            parent = x.parent
            for k, v in list(parent.children.items()):                   # radix_cache.py:412-418
                if v is x:
                    del parent.children[k]
                    break
            self.evictable -= len(x.key)
            if not parent.children:
                heapq.heappush(leaves, parent)
        return freed

    def pages_of(self, value: List[int]) -> List[int]:
        """Pages a run of page-aligned slots occupies (the correct argument for PageAllocator.free)."""
        return [value[i] // self.ps for i in range(0, len(value), self.ps)]


# ------------------------------------------------------------------------------------------ request level
class ReqState:
    """What the request-level calls read and write on a request (mini_sglang/managers/req_info.py:66-100)."""

    def __init__(self, token_ids: List[int], row: int):
        self.token_ids = list(token_ids)
        self.row = row                      # req_pool_idx
        self.prefix: List[int] = []         # prefix_indices
        self.last_node: Optional[Node] = None


def cache_unfinished_req(tree: RadixOracle, req: ReqState, table: List[List[int]], free_pages: List[int]):
    """radix_cache.py:270-332.  table[row][pos] = slot; free_pages: the allocator's list, freed pages appended."""
    ps = tree.ps
    tokens_len = len(req.token_ids) - 1
    aligned = tokens_len // ps * ps
    row = table[req.row]
    kv = row[:tokens_len]
    new_prefix_len = tree.insert(req.token_ids[:aligned], list(row[:aligned]))
    old_aligned = len(req.prefix) // ps * ps
    free_pages.extend(row[p * ps] // ps for p in range(old_aligned // ps, new_prefix_len // ps))   # our duplicates
    new_prefix, new_last = tree.match_prefix(req.token_ids[:aligned])
    row[old_aligned: len(new_prefix)] = new_prefix[old_aligned:]
    tree.dec_lock_ref(req.last_node)
    tree.inc_lock_ref(new_last)
    req.prefix = list(new_prefix) if ps == 1 else list(new_prefix) + kv[len(new_prefix):]
    req.last_node = new_last


def cache_finished_req(tree: RadixOracle, req: ReqState, table: List[List[int]], free_pages: List[int], free_rows: List[int]):
    """radix_cache.py:334-365."""
    ps = tree.ps
    tokens_len = len(req.token_ids) - 1
    aligned = tokens_len // ps * ps
    row = table[req.row]
    new_prefix_len = tree.insert(req.token_ids[:aligned], list(row[:aligned]))
    old_aligned = len(req.prefix) // ps * ps
    free_pages.extend(row[p * ps] // ps for p in range(old_aligned // ps, new_prefix_len // ps))
    if ps > 1 and tokens_len % ps != 0:
        free_pages.append(row[tokens_len // ps * ps] // ps)              # the last, incomplete page
    free_rows.append(req.row)
    tree.dec_lock_ref(req.last_node)
