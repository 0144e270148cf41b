"""Seeded random shapes through the decode and extend entry points against the oracle: lengths, page sizes, head
layouts, split sizes, launch plans and shared-prefix groups drawn together, so that combinations no hand-written
case lists get exercised (each case stays small enough for the CPU oracle)."""
import os
import random

import pytest
import torch

import test_kernels_gpu as tk

pytestmark = pytest.mark.gpu
DEV = tk.DEV
MORE = int(os.environ.get("MSGL_RANDOM_CASES", "0"))  # extra seeds per test for an occasional longer soak
HEADS = [(32, 8), (16, 8), (8, 1), (4, 1), (16, 4), (8, 2), (64, 8), (8, 8)]


@pytest.mark.parametrize("seed", range(32 + MORE))
def test_decode_random_case(seed):
    rng = random.Random(1000 + seed)
    ops = tk._ops()
    Hq, Hkv = rng.choice(HEADS)
    ps = rng.choice([1, 16, 16, 64])
    bs = rng.randint(1, 12)
    seq = [rng.choice([1, 2, 15, 16, 17, rng.randint(1, 300), rng.randint(1, 3000)]) for _ in range(bs)]
    case = tk.build_case(Hq, Hkv, ps, seq, seed=seed + 500)
    chunk = 16 * rng.randint(1, 40)
    max_splits = max(1, -(-max(seq) // chunk)) + rng.randint(0, 3)
    kw = dict(max_splits=max_splits, chunk_tokens=chunk)
    base = tk.run_decode(case, ps, **kw)
    tk.check_decode(case, base)
    if max_splits > 1:
        plan = ops.decode_plan(case["seq_lens"].to(DEV), max_splits, chunk)
        assert torch.equal(tk.run_decode(case, ps, plan=plan, **kw), base)
    cd = torch.tensor([chunk], dtype=torch.int32, device=DEV)
    assert torch.equal(tk.run_decode(case, ps, max_splits=max_splits, chunk_dev=cd), base)


@pytest.mark.parametrize("seed", range(20 + MORE))
def test_decode_shared_prefix_random_case(seed):
    rng = random.Random(2000 + seed)
    Hq, Hkv = rng.choice([(32, 8), (16, 8), (8, 1), (4, 1), (16, 4), (8, 8)])
    ps = rng.choice([1, 16, 64])
    ngroups = rng.randint(1, 4)
    prefix = [(-(-rng.randint(1, 900) // ps)) * ps for _ in range(ngroups)]
    groups = [rng.randrange(ngroups) for _ in range(rng.randint(2, 14))]
    groups = [g for g in groups] + list(range(ngroups))  # every group has a member; some only one
    rng.shuffle(groups)
    seq = [prefix[g] + rng.choice([0, 1, 5, rng.randint(1, 1200)]) for g in groups]
    seq = [max(1, s) for s in seq]
    case = tk.build_case(Hq, Hkv, ps, seq, seed=seed + 700, groups=groups, prefix=prefix)
    key_chunk = 64 * rng.randint(1, 16)
    chunk = 16 * rng.randint(2, 32)
    most_extra = max(-(-p // key_chunk) for p in prefix)
    max_splits = -(-max(seq) // chunk) + most_extra + rng.randint(0, 2)
    out = tk.run_decode_shared(case, ps, groups, prefix, key_chunk, chunk, max_splits, planned=rng.random() < 0.5)
    tk.check_decode(case, out)


@pytest.mark.parametrize("seed", range(20 + MORE))
def test_extend_random_case(seed, variants):
    rng = random.Random(3000 + seed)
    Hq, Hkv = rng.choice([(32, 8), (16, 8), (8, 1), (4, 1), (4, 4), (16, 4)])
    ps = rng.choice([1, 16, 64])
    bs = rng.randint(1, 7)
    prefix = [rng.choice([0, 0, (-(-rng.randint(1, 1500) // ps)) * ps]) for _ in range(bs)]
    inputs = [rng.choice([1, 2, 31, 32, 33, 127, 128, 129, rng.randint(1, 700)]) for _ in range(bs)]
    case = tk.build_extend_case(Hq, Hkv, ps, prefix, inputs, seed=seed + 900)
    base = tk.run_extend(case, ps)
    tk.check_extend(case, base)
    kmax = max(p + n for p, n in zip(prefix, inputs))
    # The hints (longest request, launch plan) choose among kernels and launch orders, never results: the two-slot and
    # the eight-wave kernel do the same arithmetic in the same order (equal to the last bit); the four-wave kernel, the
    # default for long or unannounced key ranges, lets a row's reference maximum lag, so it agrees to rounding.
    hinted = tk.run_extend(case, ps, max_seqlen_k=kmax, planned=True)
    tk.check_extend(case, hinted)
    assert (hinted.float() - base.float()).abs().max().item() <= 1e-2
    variants("extend_wide", "0")
    assert torch.equal(tk.run_extend(case, ps, max_seqlen_k=kmax, planned=True), tk.run_extend(case, ps))
