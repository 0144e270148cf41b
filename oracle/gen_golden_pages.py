#!/usr/bin/env python3
"""Generate tests/golden/pages.pt by driving the REFERENCE's PageAllocator (mini_sglang/mem_cache/token2kv.py:154-212)
with a seeded sequence of alloc / free / clear calls.  Run in the build container only:

    python oracle/gen_golden_pages.py [--out tests/golden] [--ref /root/reference]

The fixture holds the op sequence and the reference's observable outputs: the page ids every alloc returned, whether it
raised, and the free list (in order) after every call.  Test infrastructure: nothing in the product imports this."""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mini_sglang_amd.runtime import synth  # noqa: E402
from oracle.gen_golden import import_reference  # noqa: E402


def op_sequence(seed: int, page_num: int, n_ops: int):
    """(kind, argument) pairs: allocs of 0..page_num/3 pages (some larger than what is free), frees of a random subset
    of what is out, in a random order, and a rare clear."""
    r = synth.uniform_int(3 * n_ops, seed, 0, 1 << 20).tolist()
    return [(r[3 * i] % 16, r[3 * i + 1], r[3 * i + 2]) for i in range(n_ops)]


def run(alloc_cls, page_num: int, page_size: int, seed: int, n_ops: int = 400):
    a = alloc_cls(page_num, page_size, "cpu")
    out, held = [], []
    for kind, x, y in op_sequence(seed, page_num, n_ops):
        rec = {}
        if kind < 8:  # alloc
            num = x % (page_num // 3 + 2)
            rec["op"], rec["arg"] = "alloc", num
            try:
                got = a.alloc(num)
                rec["pages"] = [int(v) for v in got.tolist()]
                held.extend(rec["pages"])
            except RuntimeError:
                rec["raised"] = True
        elif kind < 15:  # free some of what is out, rotated so that the order differs from the allocation order
            n = x % (len(held) + 1)
            rot = y % max(1, len(held))
            held = held[rot:] + held[:rot]
            give, held = held[:n], held[n:]
            rec["op"], rec["arg"] = "free", give
            a.free(torch.tensor(give, dtype=torch.int32))
        else:
            rec["op"], rec["arg"] = "clear", 0
            a.clear()
            held = []
        rec["free_list"] = [int(v) for v in a.free_page.tolist()]
        rec["available"] = int(a.available_size())
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    import_reference(args.ref)
    from mini_sglang.mem_cache.token2kv import PageAllocator as RefAllocator

    cases = {}
    for page_num, page_size, seed in ((37, 1, 11), (64, 16, 12), (5, 64, 13)):
        cases[f"pages{page_num}_ps{page_size}"] = {"page_num": page_num, "page_size": page_size, "seed": seed,
                                                  "trace": run(RefAllocator, page_num, page_size, seed)}
    path = os.path.join(args.out, "pages.pt")
    torch.save(cases, path)
    print("wrote", path, {k: len(v["trace"]) for k, v in cases.items()})


if __name__ == "__main__":
    main()
