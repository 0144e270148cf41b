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
            if op[1] < len(recent):
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
