#!/usr/bin/env python3
"""Generate tests/golden/radix_*.pt by driving the REFERENCE's RadixCache (mini_sglang/mem_cache/radix_cache.py).

Run in the build container only:  python oracle/gen_golden_radix.py [--out tests/golden] [--ref /root/reference]

The reference module needs Python >= 3.12 for one nested-quote f-string inside its debug printer
(_print_tree, radix_cache.py:243); under this container's 3.10 it is loaded from its source text with that one
expression's inner quotes changed to single quotes — the printer is never called here — and everything else
(tree, split, lock counting, eviction heap, request caching) runs as the reference wrote it, on its own
ReqToTokenPool / PageAllocator / Req classes (CPU tensors).  Fixtures hold only the op sequences (seeds) and
the reference's observable OUTPUTS.

  radix_tree.pt   per page size: a seeded sequence of insert / match_prefix / lock / unlock / evict calls with every
                  return value, the evictable / protected sizes after each call and the values handed to the allocator
  radix_reqs.pt   page_size 1: requests that share prefixes go through match_prefix -> lock -> (extend slots written)
                  -> cache_unfinished_req -> (decode slots) -> cache_finished_req, with eviction under pressure:
                  the page-table rows, prefix lengths, free list and sizes after every step
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mini_sglang_amd.runtime import synth  # noqa: E402
from oracle.gen_golden import import_reference  # noqa: E402


def load_reference_radix(ref_root: str):
    path = os.path.join(ref_root, "mini_sglang", "mem_cache", "radix_cache.py")
    src = open(path).read()
    bad = 'f"{"  " * depth}: {len(node.key)}, {node.key[:10]}, r={node.lock_ref}"'
    assert src.count(bad) == 1, "reference source changed: re-check the one-line syntax adaptation"
    src = src.replace(bad, "f\"{'  ' * depth}: {len(node.key)}, {node.key[:10]}, r={node.lock_ref}\"")
    mod = types.ModuleType("mini_sglang.mem_cache.radix_cache")
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def digest(t) -> tuple:
    """(length, position-weighted checksum) of an int tensor / list: what the fixtures keep instead of the values."""
    t = torch.as_tensor(t, dtype=torch.int64).reshape(-1)
    w = torch.arange(1, t.numel() + 1, dtype=torch.int64)
    return (int(t.numel()), int((t * w).sum() % 2147483647))


# ------------------------------------------------------------------------------------------ op sequences
def documents(seed: int, n_docs: int, length: int, vocab: int = 50):
    return [synth.uniform_int(length, seed + d, 0, vocab).tolist() for d in range(n_docs)]


def tree_ops(seed: int, page_size: int, n_ops: int = 500):
    """A seeded op list: keys are prefixes of a few documents with occasional divergence, so that inserts split
    nodes, matches stop inside nodes, and leaves of different ages exist for eviction."""
    ps = page_size
    docs = documents(seed, 6, 40 * ps)
    r = synth.uniform_int(n_ops * 6, seed + 99, 0, 1 << 30).tolist()
    ops, next_base = [], 1000
    for i in range(n_ops):
        a, b, c, d, e, f = r[6 * i: 6 * i + 6]
        doc = list(docs[a % len(docs)])
        n = 1 + b % len(doc)
        key = doc[:n]
        if c % 3 == 0 and n > 2:                      # diverge from the document somewhere inside
            p = d % n
            key = key[:p] + [(t + 1 + e % 7) % 50 for t in key[p:]]
        kind = f % 10
        if kind < 4:
            key = key[: len(key) // ps * ps]          # callers insert page-aligned keys (radix_cache.py:286, 343)
            ops.append(("insert", key, next_base))
            next_base += len(key) + 3
        elif kind < 7:
            ops.append(("match", key))
        elif kind == 7:
            ops.append(("lock", a % 8))               # lock / unlock the node of the j-th most recent match
        elif kind == 8:
            ops.append(("unlock", a % 8))
        else:
            ops.append(("evict", (1 + b % 30) * ps))
    return ops


def attached(tree, node) -> bool:
    """True while `node` is still part of the tree (eviction detaches nodes; a scheduler never locks those)."""
    while node is not tree.root:
        parent = node.parent
        if parent is None or not any(c is node for c in parent.children.values()):
            return False
        node = parent
    return True


def run_tree_ops(RadixCache, ops, page_size):
    freed = []

    class Recorder:  # stands in for the page allocator: evict hands it the evicted values (radix_cache.py:383-384)
        def free(self, t):
            freed.append(t.clone())

    tree = RadixCache(None, None, Recorder(), page_size)
    recent, locked, out = [], [], []
    for op in ops:
        rec = {"op": op[0]}
        if op[0] == "insert":
            key, base = op[1], op[2]
            rec["ret"] = int(tree.insert(list(key), torch.arange(base, base + len(key), dtype=torch.int32)))
        elif op[0] == "match":
            res = tree.match_prefix(list(op[1]))
            rec["indices"] = digest(res.match_indices)
            recent.insert(0, res.last_node)
            del recent[8:]
        elif op[0] == "lock":
            rec["done"] = op[1] < len(recent) and attached(tree, recent[op[1]])
            if rec["done"]:
                tree.inc_lock_ref(recent[op[1]])
                locked.append(recent[op[1]])
        elif op[0] == "unlock":
            if locked:
                tree.dec_lock_ref(locked.pop(op[1] % len(locked)))
        elif op[0] == "evict":
            n0 = len(freed)
            tree.evict(op[1])
            rec["freed"] = [digest(t) for t in freed[n0:]]   # one entry per evicted node, in eviction order
        rec["evictable"], rec["protected"] = int(tree.evictable_size()), int(tree.protected_size())
        out.append(rec)
    return out


# ------------------------------------------------------------------------------------------ request level
def req_scenario(seed: int, n_reqs: int = 24):
    """(prompt tokens, new tokens) per request: prompts share document prefixes."""
    docs = documents(seed, 4, 60)
    r = synth.uniform_int(n_reqs * 4, seed + 7, 0, 1 << 30).tolist()
    reqs = []
    for i in range(n_reqs):
        a, b, c, d = r[4 * i: 4 * i + 4]
        n = 4 + b % 50
        prompt = docs[a % 4][:n] + [50 + (c + j) % 20 for j in range(c % 6)]
        reqs.append((prompt, 1 + d % 12))
    return reqs


def run_req_scenario(mods, scenario, pool_pages: int):
    """The scheduler's sequence for one request at a time (scheduler.py:326-349, 421-432; batch_info.py:175-235):
    calc_prefix, lock, allocate (evicting when short), write the table row, cache_unfinished_req after the prefill,
    decode steps appending one slot each, cache_finished_req."""
    RadixCache = mods["radix"].RadixCache
    from mini_sglang.managers.req_info import Req
    from mini_sglang.managers.sampling_params import SamplingParams
    from mini_sglang.mem_cache.req2token import ReqToTokenPool
    from mini_sglang.mem_cache.token2kv import PageAllocator

    r2t = ReqToTokenPool(8, 256, 1, "cpu")
    pa = PageAllocator(pool_pages, 1, "cpu")
    tree = RadixCache(r2t, None, pa, 1)
    out = []
    for i, (prompt, new) in enumerate(scenario):
        req = Req(str(i), list(prompt), SamplingParams())
        req.calc_prefix(tree)                                    # matches token_ids[:-1] (req_info.py:142-148)
        tree.inc_lock_ref(req.last_node)
        prefix = req.prefix_indices.clone()
        need = len(prompt) - len(prefix)
        if pa.available_size() < need:
            tree.evict(need - pa.available_size())
        slots = pa.alloc(need)
        req.req_pool_idx = r2t.alloc(1)[0]
        r2t.write_tokens(req.req_pool_idx, slice(0, len(prefix)), prefix)
        r2t.write_tokens(req.req_pool_idx, slice(len(prefix), len(prompt)), slots)
        req.token_ids.append(7)                                  # the prefill's sampled token
        tree.cache_unfinished_req(req)
        rec = {"prefix_len": int(len(prefix)), "first_slots": digest(slots),
               "after_unfinished": (int(tree.evictable_size()), int(tree.protected_size()))}
        for j in range(new - 1):                                 # decode steps: one slot per step
            if pa.available_size() < 1:
                tree.evict(1)
            s = pa.alloc(1)
            r2t.write_tokens(req.req_pool_idx, slice(len(req.token_ids) - 1, len(req.token_ids)), s)
            req.token_ids.append(8 + j)
        rec["row"] = digest(r2t.req_to_token[req.req_pool_idx, : len(req.token_ids) - 1])
        tree.cache_finished_req(req)
        rec["after_finished"] = (int(tree.evictable_size()), int(tree.protected_size()))
        rec["free_pages"] = digest(pa.free_page)
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    import_reference(a.ref)
    radix = load_reference_radix(a.ref)
    fx = {}
    for ps in (1, 16):
        fx[f"ps{ps}"] = run_tree_ops(radix.RadixCache, tree_ops(5 + ps, ps), ps)
    torch.save(fx, os.path.join(a.out, "radix_tree.pt"))
    print("radix_tree.pt:", {k: len(v) for k, v in fx.items()})
    rq = run_req_scenario({"radix": radix}, req_scenario(11), pool_pages=160)
    torch.save({"steps": rq}, os.path.join(a.out, "radix_reqs.pt"))
    print("radix_reqs.pt:", len(rq), "requests; sizes", os.path.getsize(os.path.join(a.out, "radix_tree.pt")),
          os.path.getsize(os.path.join(a.out, "radix_reqs.pt")))


if __name__ == "__main__":
    main()
