"""The page allocator of libmsgl_host.so (C++ ring buffer behind the reference's PageAllocator interface) against the
REFERENCE class's own behaviour, recorded in tests/golden/pages.pt by oracle/gen_golden_pages.py: 3 x 400 seeded
alloc / free / clear calls; every returned page list, every refusal and the free list after every call must agree.
The order of the free list is the contract: it decides every slot a request is given."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mini_sglang_amd.mem_cache import PageAllocator  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "pages.pt")


@pytest.mark.parametrize("case", ["pages37_ps1", "pages64_ps16", "pages5_ps64"])
def test_allocator_matches_reference_trace(case):
    rec = torch.load(GOLDEN, weights_only=True)[case]
    a = PageAllocator(rec["page_num"], rec["page_size"])
    assert a.free_page.tolist() == list(range(1, rec["page_num"] + 1))
    for i, step in enumerate(rec["trace"]):
        if step["op"] == "alloc":
            if step.get("raised"):
                before = a.free_page.tolist()
                with pytest.raises(RuntimeError, match="Not enough free pages"):
                    a.alloc(step["arg"])
                assert a.free_page.tolist() == before, "a refused alloc must take nothing"
            else:
                got = a.alloc(step["arg"])
                assert got.dtype == torch.int32 and got.tolist() == step["pages"], f"step {i}"
        elif step["op"] == "free":
            a.free(torch.tensor(step["arg"], dtype=torch.int32))
        else:
            a.clear()
        assert a.free_page.tolist() == step["free_list"], f"free list after step {i} ({step['op']})"
        assert a.available_size() == step["available"]


def test_allocator_edges():
    a = PageAllocator(4, 16)
    assert a.alloc(0).numel() == 0 and a.free_count() == 4
    got = a.alloc(4)
    assert got.tolist() == [1, 2, 3, 4] and a.available_size() == 0
    with pytest.raises(RuntimeError):
        a.alloc(1)
    with pytest.raises(ValueError):
        a.free(torch.tensor([0], dtype=torch.int32))  # page 0 is the dummy page, never handed out
    with pytest.raises(ValueError):
        a.free(torch.tensor([5], dtype=torch.int32))
    assert a.free_count() == 0, "a refused free must add nothing"
    a.free(got.flip(0).to(torch.int64))  # any integer dtype, any order
    assert a.free_page.tolist() == [4, 3, 2, 1]
    # freeing a page twice is the caller's bug (the reference does not notice either); it must not corrupt the list
    a.free(torch.tensor([2, 2, 2, 2, 2, 2, 2], dtype=torch.int32))
    assert a.free_page.tolist() == [4, 3, 2, 1] + [2] * 7 and a.alloc(5).tolist() == [4, 3, 2, 1, 2]
    # a pool laid out by hand, as the synthetic workloads do
    a.free_page = torch.arange(3, 5, dtype=torch.int32)
    assert a.free_page.tolist() == [3, 4] and a.alloc(1).tolist() == [3]
    a.clear()
    assert a.free_page.tolist() == [1, 2, 3, 4]
    # long runs of alloc / free wrap around the ring many times
    b = PageAllocator(7, 1)
    held = []
    for i in range(1000):
        held += b.alloc(3).tolist()
        b.free(torch.tensor(held[:2], dtype=torch.int32))
        held = held[2:]
        if len(held) > 4:
            b.free(torch.tensor(held, dtype=torch.int32))
            held = []
    assert sorted(b.free_page.tolist() + held) == list(range(1, 8))
