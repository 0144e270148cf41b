"""Radix prefix cache: the CPU restatement (oracle/radix_oracle.py) pinned to the reference's own RadixCache
through tests/golden/radix_tree.pt and radix_reqs.pt, and the C++ tree (libmsgl_host.so behind
mini_sglang_amd.mem_cache.radix_cache.RadixCache) checked against the oracle and the same fixtures.  Host code
only: runs on the CPU."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import golden_cases as gc  # noqa: E402
from oracle import gen_golden_radix as gr  # noqa: E402
from oracle import radix_oracle as ro  # noqa: E402


class OracleDriver:
    """Adapter: the fixture's op vocabulary on the oracle tree."""

    def __init__(self, ps):
        self.tree = ro.RadixOracle(ps)

    def insert(self, key, base):
        return self.tree.insert(list(key), list(range(base, base + len(key))))

    def match(self, key):
        return self.tree.match_prefix(list(key))

    def lock(self, node):
        self.tree.inc_lock_ref(node)

    def unlock(self, node):
        self.tree.dec_lock_ref(node)

    def evict(self, n):
        return self.tree.evict(n)          # value lists of the evicted nodes, eviction order

    def sizes(self):
        return self.tree.evictable, self.tree.protected


def replay_tree_ops(driver, ops, want):
    recent, locked = [], []
    for i, (op, rec) in enumerate(zip(ops, want)):
        assert rec["op"] == op[0]
        if op[0] == "insert":
            assert driver.insert(op[1], op[2]) == rec["ret"], f"op {i}"
        elif op[0] == "match":
            idx, node = driver.match(op[1])
            assert gr.digest(idx) == tuple(rec["indices"]), f"op {i}"
            recent.insert(0, node)
            del recent[8:]
        elif op[0] == "lock":
            if rec["done"]:  # the generator skips handles whose node has been evicted since the match
                driver.lock(recent[op[1]])
                locked.append(recent[op[1]])
        elif op[0] == "unlock":
            if locked:
                driver.unlock(locked.pop(op[1] % len(locked)))
        elif op[0] == "evict":
            freed = driver.evict(op[1])
            assert [gr.digest(v) for v in freed] == [tuple(d) for d in rec["freed"]], f"op {i}"
        assert driver.sizes() == (rec["evictable"], rec["protected"]), f"op {i} ({op[0]})"


@pytest.mark.parametrize("ps", [1, 16])
def test_oracle_tree_matches_reference(ps):
    fx = gc.load("radix_tree.pt")[f"ps{ps}"]
    replay_tree_ops(OracleDriver(ps), gr.tree_ops(5 + ps, ps), fx)


def run_req_scenario_oracle(scenario, pool_pages):
    tree = ro.RadixOracle(1)
    table = [[0] * 256 for _ in range(8)]
    free_rows = list(range(8))
    free_pages = list(range(1, pool_pages + 1))
    out = []
    for prompt, new in scenario:
        req = ro.ReqState(prompt, -1)
        req.prefix, req.last_node = tree.match_prefix(req.token_ids[:-1])
        tree.inc_lock_ref(req.last_node)
        prefix = list(req.prefix)
        need = len(prompt) - len(prefix)
        if len(free_pages) < need:
            for v in tree.evict(need - len(free_pages)):
                free_pages.extend(tree.pages_of(v))
        slots, free_pages[:] = free_pages[:need], free_pages[need:]
        req.row = free_rows.pop(0)
        table[req.row][: len(prefix)] = prefix
        table[req.row][len(prefix): len(prompt)] = slots
        req.token_ids.append(7)
        ro.cache_unfinished_req(tree, req, table, free_pages)
        rec = {"prefix_len": len(prefix), "first_slots": gr.digest(slots), "after_unfinished": (tree.evictable, tree.protected)}
        for j in range(new - 1):
            if len(free_pages) < 1:
                for v in tree.evict(1):
                    free_pages.extend(tree.pages_of(v))
            s = free_pages.pop(0)
            table[req.row][len(req.token_ids) - 1] = s
            req.token_ids.append(8 + j)
        rec["row"] = gr.digest(table[req.row][: len(req.token_ids) - 1])
        ro.cache_finished_req(tree, req, table, free_pages, free_rows)
        rec["after_finished"] = (tree.evictable, tree.protected)
        rec["free_pages"] = gr.digest(free_pages)
        out.append(rec)
    return out


def test_oracle_request_caching_matches_reference():
    fx = gc.load("radix_reqs.pt")["steps"]
    got = run_req_scenario_oracle(gr.req_scenario(11), pool_pages=160)
    assert any(r["prefix_len"] > 0 for r in fx), "the scenario must exercise prefix hits"
    for i, (g, w) in enumerate(zip(got, fx)):
        for k in w:
            assert tuple(g[k]) == tuple(w[k]) if isinstance(w[k], (tuple, list)) else g[k] == w[k], f"request {i}: {k}"


# ------------------------------------------------------------------------------------------ the C++ tree
class NativeDriver:
    """The same op vocabulary on mini_sglang_amd.mem_cache.radix_cache.RadixCache (C++ tree through the C ABI)."""

    def __init__(self, ps):
        from mini_sglang_amd.mem_cache.radix_cache import RadixCache
        self.tree = RadixCache(None, None, None, ps)

    def insert(self, key, base):
        return self.tree.insert(list(key), torch.arange(base, base + len(key), dtype=torch.int32))

    def match(self, key):
        res = self.tree.match_prefix(list(key))
        return res.match_indices.tolist(), res.last_node

    def lock(self, node):
        self.tree.inc_lock_ref(node)

    def unlock(self, node):
        self.tree.dec_lock_ref(node)

    def evict(self, n):
        return [v.tolist() for v in self.tree.evict_nodes(n)]

    def sizes(self):
        return self.tree.evictable_size(), self.tree.protected_size()


def test_host_library_exports_every_declared_symbol():
    import re
    from mini_sglang_amd import _capi_host
    header = open(os.path.join(ROOT, "include", "msgl_host.h")).read()
    declared = set(re.findall(r"\b(msgl_(?:radix|pages)_\w+)\s*\(", header))
    assert declared == set(_capi_host.SIGNATURES), declared ^ set(_capi_host.SIGNATURES)
    for name in declared:
        assert hasattr(_capi_host.lib, name)


@pytest.mark.parametrize("ps", [1, 16])
def test_native_tree_matches_reference(ps):
    fx = gc.load("radix_tree.pt")[f"ps{ps}"]
    replay_tree_ops(NativeDriver(ps), gr.tree_ops(5 + ps, ps), fx)


@pytest.mark.parametrize("ps,seed", [(1, 101), (4, 102), (16, 103), (64, 104)])
def test_native_tree_matches_oracle_on_other_sequences(ps, seed):
    """Longer op sequences than the fixtures hold, oracle and C++ tree side by side (results, sizes, eviction order)."""
    ops = gr.tree_ops(seed, ps, n_ops=1500)
    a, b = OracleDriver(ps), NativeDriver(ps)
    ra, rb, la, lb = [], [], [], []
    for i, op in enumerate(ops):
        if op[0] == "insert":
            assert a.insert(op[1], op[2]) == b.insert(op[1], op[2]), f"op {i}"
        elif op[0] == "match":
            (ia, na), (ib, nb) = a.match(op[1]), b.match(op[1])
            assert ia == ib, f"op {i}"
            ra.insert(0, na), rb.insert(0, nb)
            del ra[8:], rb[8:]
        elif op[0] == "lock":
            if op[1] < len(ra):
                if gr.attached(a.tree, ra[op[1]]):
                    a.lock(ra[op[1]]), b.lock(rb[op[1]])
                    la.append(ra[op[1]]), lb.append(rb[op[1]])
                else:  # evicted since the match: the C++ tree has dropped the handle
                    with pytest.raises(KeyError):
                        b.lock(rb[op[1]])
        elif op[0] == "unlock":
            if la:
                j = op[1] % len(la)
                a.unlock(la.pop(j)), b.unlock(lb.pop(j))
        elif op[0] == "evict":
            assert a.evict(op[1]) == b.evict(op[1]), f"op {i}"
        assert a.sizes() == b.sizes(), f"op {i}"


def test_native_tree_edge_cases():
    from mini_sglang_amd.mem_cache.radix_cache import RadixCache
    t = RadixCache(None, None, None, 4)
    assert t.match_prefix([]).match_indices.numel() == 0 and t.match_prefix([]).last_node == 0
    assert t.match_prefix([1, 2, 3]).match_indices.numel() == 0          # shorter than a page
    assert t.insert([], []) == 0
    assert t.insert([1, 2, 3, 4, 5, 6, 7, 8], list(range(10, 18))) == 0
    assert t.match_prefix([1, 2, 3, 4, 9, 9, 9, 9]).match_indices.tolist() == [10, 11, 12, 13]   # splits the node
    assert t.match_prefix([1, 2, 3, 4, 5, 6, 7, 8, 1]).match_indices.tolist() == list(range(10, 18))
    assert t.evictable_size() == 8 and t.protected_size() == 0
    node = t.match_prefix([1, 2, 3, 4, 5, 6, 7, 8]).last_node
    t.inc_lock_ref(node)
    assert (t.evictable_size(), t.protected_size()) == (0, 8)
    assert t.evict_nodes(100) == []                                      # everything is locked
    t.dec_lock_ref(node)
    assert [v.tolist() for v in t.evict_nodes(1)] == [[14, 15, 16, 17]]  # the leaf goes first, a whole node at a time
    with pytest.raises(KeyError):
        t.inc_lock_ref(node)                                             # that node is gone
    t.reset()
    assert t.evictable_size() == 0 and t.match_prefix([1, 2, 3, 4]).match_indices.numel() == 0


def test_native_request_caching_matches_reference():
    """The scheduler's per-request sequence on this repo's pools, allocator and RadixCache mirror reproduces the
    reference's page-table rows, free list and sizes (fixture radix_reqs.pt, page_size 1)."""
    from mini_sglang_amd.mem_cache import PageAllocator, ReqToTokenPool
    from mini_sglang_amd.mem_cache.radix_cache import RadixCache
    from mini_sglang_amd.runtime.batch_info import Req

    fx = gc.load("radix_reqs.pt")["steps"]
    r2t = ReqToTokenPool(8, 256, 1, "cpu")
    pa = PageAllocator(160, 1)
    tree = RadixCache(r2t, None, pa, 1)
    for i, ((prompt, new), want) in enumerate(zip(gr.req_scenario(11), fx)):
        req = Req(i, list(prompt))
        m = tree.match_prefix(req.token_ids[:-1])
        req.prefix_indices, req.last_node = m.match_indices, m.last_node
        tree.inc_lock_ref(req.last_node)
        prefix = req.prefix_indices.clone()
        need = len(prompt) - prefix.numel()
        if pa.available_size() < need:
            tree.evict(need - pa.available_size())
        slots = pa.alloc(need)
        req.req_pool_idx = r2t.alloc(1)[0]
        r2t.req_to_token[req.req_pool_idx, : prefix.numel()] = prefix
        r2t.req_to_token[req.req_pool_idx, prefix.numel(): len(prompt)] = slots
        req.token_ids.append(7)
        tree.cache_unfinished_req(req)
        assert prefix.numel() == want["prefix_len"] and gr.digest(slots) == tuple(want["first_slots"]), f"request {i}"
        assert (tree.evictable_size(), tree.protected_size()) == tuple(want["after_unfinished"]), f"request {i}"
        for j in range(new - 1):
            if pa.available_size() < 1:
                tree.evict(1)
            s = pa.alloc(1)
            r2t.req_to_token[req.req_pool_idx, len(req.token_ids) - 1] = s
            req.token_ids.append(8 + j)
        assert gr.digest(r2t.req_to_token[req.req_pool_idx, : len(req.token_ids) - 1]) == tuple(want["row"]), f"request {i}"
        tree.cache_finished_req(req)
        assert (tree.evictable_size(), tree.protected_size()) == tuple(want["after_finished"]), f"request {i}"
        assert gr.digest(pa.free_page) == tuple(want["free_pages"]), f"request {i}"
