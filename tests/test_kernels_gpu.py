"""Parity of every HIP kernel (called through the C ABI via mini_sglang_amd.ops) against the CPU oracle on
seeded inputs.  Tolerances: attention outputs 2e-2 abs vs the oracle (the bound the north star puts on
logits) and 1e-2 abs vs an fp64 evaluation; layer kernels at most one bf16 ulp apart from the oracle on a
small fraction of elements, bit-exact for RoPE in reference-rounding mode and for integer/byte work."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as orc  # noqa: E402


def _ops():
    from mini_sglang_amd import ops
    return ops


def _synth():
    from mini_sglang_amd.runtime import synth, workload
    return synth, workload


DEV = "cuda"


def ulp_diff_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 ulps (monotone integer mapping of the bit patterns)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a.cpu()) - key(b.cpu())).abs()


def assert_close_fp64(out: torch.Tensor, exact: torch.Tensor, what: str, slack: float = 1.0):
    """bf16 output vs an fp64 evaluation: one bf16 ulp (2^-8 relative) plus 4e-3 absolute, times `slack`."""
    err = (out.double() - exact).abs()
    bound = slack * (4e-3 + exact.abs() * 2.0 ** -8)
    worst = (err - bound).max().item()
    assert worst <= 0, f"{what} vs fp64: exceeds bound by {worst} (max abs err {err.max().item()})"


def build_case(Hq, Hkv, page_size, seq_lens, seed, groups=None, prefix=None, layout="interleaved"):
    synth, workload = _synth()
    D = 128
    bs = len(seq_lens)
    pool_pages = sum(-(-s // page_size) for s in seq_lens) + 8
    pt = workload.make_page_table(seq_lens, page_size, pool_pages, seed, groups=groups, group_prefix_lens=prefix)
    slots = (pool_pages + 1) * page_size
    if layout == "interleaved":
        kv = synth.normal((slots, 2, Hkv, D), seed + 1)
        k_cache, v_cache = kv[:, 0], kv[:, 1]
    else:
        k_cache, v_cache = synth.normal((slots, Hkv, D), seed + 1), synth.normal((slots, Hkv, D), seed + 2)
    q = synth.normal((bs, Hq, D), seed + 3)
    # more requests rows than batch entries: request b lives in row perm[b]
    rows = bs + 3
    perm = synth.permutation(rows, seed + 4)[:bs].to(torch.int32)
    r2t = torch.zeros((rows, pt.req_to_token.shape[1]), dtype=torch.int32)
    r2t[perm.long()] = pt.req_to_token
    return dict(q=q, k_cache=k_cache, v_cache=v_cache, r2t=r2t, req_idx=perm, seq_lens=pt.seq_lens, kv_full=kv if layout == "interleaved" else None)


def run_decode(case, page_size, **kw):
    ops = _ops()
    if case["kv_full"] is not None:
        kvd = case["kv_full"].to(DEV)
        kc, vc = kvd[:, 0], kvd[:, 1]
    else:
        kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    out = ops.paged_decode(case["q"].to(DEV), kc, vc, case["r2t"].to(DEV), case["req_idx"].to(DEV),
                           case["seq_lens"].to(DEV), 128 ** -0.5, page_size, **kw)
    torch.cuda.synchronize()
    return out.cpu()


def check_decode(case, out):
    scale = 128 ** -0.5
    ref = orc.paged_decode_ref(case["q"], case["k_cache"], case["v_cache"], case["r2t"], case["req_idx"], case["seq_lens"], scale)
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= 2e-2, f"vs oracle: max abs err {err}"
    for b in range(len(case["seq_lens"])):
        S = int(case["seq_lens"][b])
        tok = case["r2t"][int(case["req_idx"][b]), :S]
        exact = orc.paged_attention_fp64(case["q"][b: b + 1], case["k_cache"], case["v_cache"], tok, scale)[0]
        assert_close_fp64(out[b], exact, f"request {b} (S={S})")


@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (64, 8), (8, 1), (4, 1), (16, 4), (8, 2), (8, 8), (2, 2), (1, 1)])
@pytest.mark.parametrize("page_size", [1, 16])
def test_decode_heads(Hq, Hkv, page_size):
    seq = [1, 2, 15, 16, 17, 31, 33, 100, 257, 1000]
    case = build_case(Hq, Hkv, page_size, seq, seed=10 + Hq + Hkv)
    check_decode(case, run_decode(case, page_size))


@pytest.mark.parametrize("page_size", [1, 4, 16, 48, 64])
@pytest.mark.parametrize("layout", ["interleaved", "split"])
def test_decode_page_sizes_and_splits(page_size, layout):
    ps = page_size
    seq = [1, max(1, ps - 1), ps, ps + 1, 3 * ps + 5, 1023, 1024, 1025, 2500]
    case = build_case(32, 8, ps, seq, seed=77 + ps, layout=layout)
    base = run_decode(case, ps)
    check_decode(case, base)
    for max_splits, chunk in [(4, 1024), (16, 256), (64, 48), (3, 16)]:
        out = run_decode(case, ps, max_splits=max_splits, chunk_tokens=chunk)
        check_decode(case, out)
    # chunk read from device memory (graph replay path), deliberately not a multiple of 16
    cd = torch.tensor([100], dtype=torch.int32, device=DEV)
    check_decode(case, run_decode(case, ps, max_splits=32, chunk_dev=cd))


def test_decode_shared_prefix_rows():
    ps = 16
    groups = [0, 0, 0, 1, 1, 2]
    prefix = [256, 64, 0]
    seq = [300, 256, 1000, 64, 65, 40]
    case = build_case(32, 8, ps, seq, seed=5, groups=groups, prefix=prefix)
    # rows of a group really share slots
    r2t, idx = case["r2t"], case["req_idx"].long()
    assert torch.equal(r2t[idx[0], :256], r2t[idx[1], :256]) and torch.equal(r2t[idx[0], :256], r2t[idx[2], :256])
    base = run_decode(case, ps, max_splits=8, chunk_tokens=128)
    check_decode(case, base)
    # the cache-policy hint never changes results: all-cached, all-streamed, streamed beyond the shared prefix
    reuse = torch.tensor([prefix[g] for g in groups], dtype=torch.int32, device=DEV)
    for kw in (dict(kv_stream=False), dict(kv_stream=True), dict(kv_stream=True, reuse_lens=reuse)):
        assert torch.equal(run_decode(case, ps, max_splits=8, chunk_tokens=128, **kw), base)


def run_decode_shared(case, page_size, groups, prefix, key_chunk, chunk, max_splits, planned=False):
    """Shared prefixes once per group (msgl_shared_prefix_decode), the rest per request, one merge."""
    ops = _ops()
    kvd = case["kv_full"].to(DEV)
    kc, vc = kvd[:, 0], kvd[:, 1]
    bs = len(groups)
    ng = max(groups) + 1
    members = [[i for i, g in enumerate(groups) if g == gi] for gi in range(ng)]  # anywhere in the batch
    count = [len(m) for m in members]
    first = [sum(count[:g]) for g in range(ng)]
    consecutive = all(m == list(range(m[0], m[0] + len(m))) for m in members)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    # a group of one has nothing to share: it stays with the per-request kernel
    gprefix = [prefix[g] if count[g] > 1 else 0 for g in range(ng)]
    start = i32([gprefix[g] for g in groups])
    extra = i32([-(-gprefix[g] // key_chunk) for g in groups])
    q, r2t, idx, sl = case["q"].to(DEV), case["r2t"].to(DEV), case["req_idx"].to(DEV), case["seq_lens"].to(DEV)
    ws = torch.zeros(ops.decode_workspace_bytes(bs, q.shape[1], 128, max_splits), dtype=torch.uint8, device=DEV)
    # consecutive members may be passed without the member list (group_first = batch index of the first member)
    ops.shared_prefix_decode(q, kc, vc, r2t, idx, sl, start, i32([m[0] for m in members] if consecutive else first), i32(count),
                             i32(gprefix), max(count), max(max(gprefix), 1), key_chunk, 128 ** -0.5, ws, max_splits, chunk,
                             group_members=None if consecutive else i32([i for m in members for i in m]))
    plan = ops.decode_plan(sl, max_splits, chunk, start_lens=start) if planned else None
    out = ops.paged_decode(q, kc, vc, r2t, idx, sl, 128 ** -0.5, page_size, workspace=ws, max_splits=max_splits, chunk_tokens=chunk,
                           start_lens=start, extra_splits=extra, plan=plan)
    torch.cuda.synchronize()
    return out.cpu()


def test_decode_shared_prefix_members_anywhere_in_the_batch():
    """The members of a group need not be neighbours: three groups interleaved through the batch."""
    ps = 16
    groups = [0, 1, 0, 2, 1, 0, 1, 2, 0]
    prefix = [512, 64, 1040]
    seq = [prefix[g] + 1 + 37 * i for i, g in enumerate(groups)]
    case = build_case(32, 8, ps, seq, seed=83, groups=groups, prefix=prefix)
    base = run_decode(case, ps, max_splits=16, chunk_tokens=256)
    check_decode(case, base)
    out = run_decode_shared(case, ps, groups, prefix, key_chunk=512, chunk=256, max_splits=16, planned=True)
    check_decode(case, out)
    assert (out.float() - base.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("Hq,Hkv,page_size", [(32, 8, 16), (8, 1, 64), (16, 8, 1), (4, 4, 16)])
def test_decode_shared_prefix_computed_once_per_group(Hq, Hkv, page_size):
    """Groups of requests whose page-table rows start with the same slots: the shared part goes through the
    matrix-core kernel once per group, the private part through the gather kernel, and the merged result must match
    the oracle's per-request attention over the whole row (and the plain decode kernel within rounding).  Covers a
    group wider than one 128-row tile, a member whose row is nothing but the prefix, a group of one, prefixes that
    are not multiples of the 64-key tile (page_size 1) and several key slices per prefix."""
    ps = page_size
    al = lambda n: -(-n // ps) * ps  # prefixes are page aligned
    groups = [0] * 3 + [1] * 40 + [2] + [3] * 2
    prefix = [al(300), al(1000), al(128), al(70)]
    seq = [prefix[0] + 5, prefix[0], prefix[0] + 700] + [prefix[1] + 1 + 13 * i for i in range(40)] + [500] + [prefix[3] + 1, prefix[3] + 2]
    case = build_case(Hq, Hkv, ps, seq, seed=71 + Hq, groups=groups, prefix=prefix)
    base = run_decode(case, ps, max_splits=16, chunk_tokens=256)
    check_decode(case, base)
    for key_chunk, planned in [(256, False), (1024, True)]:
        out = run_decode_shared(case, ps, groups, prefix, key_chunk, chunk=256, max_splits=16, planned=planned)
        check_decode(case, out)
        assert (out.float() - base.float()).abs().max().item() <= 2e-2


def test_decode_full_size_properties():
    """The headline shape (batch 256 x 8192 tokens, page_size 16, 32 q / 8 kv heads; 8.6 GB of K/V) through
    properties that do not need the CPU oracle for the whole batch: the split size does not change the result
    beyond rounding, permuting the requests permutes the outputs bit for bit, and sampled requests agree with
    an fp64 evaluation."""
    ops = _ops()
    synth, workload = _synth()
    bs, S, ps, Hq, Hkv, D = 256, 8192, 16, 32, 8, 128
    pages = bs * (S // ps) + 8
    pt = workload.make_page_table([S] * bs, ps, pages, seed=9)
    torch.manual_seed(91)
    kv = torch.empty(((pages + 1) * ps, 2, Hkv, D), dtype=torch.bfloat16, device=DEV).normal_()
    q = synth.normal((bs, Hq, D), 92, device=DEV)
    r2t, lens = pt.req_to_token.to(DEV), pt.seq_lens.to(DEV)
    idx = torch.arange(bs, dtype=torch.int32, device=DEV)
    run = lambda qq, ii, **kw: ops.paged_decode(qq, kv[:, 0], kv[:, 1], r2t, ii, lens, D ** -0.5, ps, **kw)
    a = run(q, idx, max_splits=2, chunk_tokens=4096)
    b = run(q, idx, max_splits=16, chunk_tokens=512)
    assert (a.float() - b.float()).abs().max().item() <= 2e-2
    perm = synth.permutation(bs, 93).to(DEV)
    c = run(q[perm].contiguous(), idx[perm].contiguous(), max_splits=2, chunk_tokens=4096)
    assert torch.equal(c, a[perm])
    for r in (0, 100, 255):
        rows = kv[r2t[r, :S].long()].cpu()  # the request's K/V rows in token order
        exact = orc.paged_attention_fp64(q[r: r + 1].cpu(), rows[:, 0], rows[:, 1], torch.arange(S), D ** -0.5)[0]
        assert_close_fp64(a[r].cpu(), exact, f"request {r} of the headline batch")


def test_decode_plan_orders_items():
    """The device-built launch order of a ragged decode batch against its restatement: same items, non-increasing
    tile counts, -1 tail; chunk from a host value and from device memory; the run-time switch to the identity."""
    ops = _ops()
    for lens, max_splits, chunk in [([5000, 100, 900, 33, 2048, 1, 4097, 16, 17], 16, 512),
                                    ([1], 1, 16), ([70000, 3], 8, 4096),   # the last split takes a long tail
                                    ([1 + (37 * i) % 3000 for i in range(1500)], 4, 1024)]:
        sl = torch.tensor(lens, dtype=torch.int32, device=DEV)
        items, tiles = orc.decode_plan_ref(lens, max_splits, chunk)
        cost = dict(zip(items, tiles))
        for kw in (dict(chunk_tokens=chunk), dict(chunk_dev=torch.tensor([chunk], dtype=torch.int32, device=DEV))):
            plan = ops.decode_plan(sl, max_splits, **kw).cpu()
            assert plan.numel() == len(lens) * max_splits and (plan[len(items):] == -1).all()
            got = plan[: len(items)].tolist()
            assert sorted(got) == sorted(items)
            got_cost = [min(cost[i], 2047) for i in got]
            assert got_cost == sorted(got_cost, reverse=True)
        off = torch.zeros(1, dtype=torch.int32, device=DEV)
        ident = ops.decode_plan(sl, max_splits, chunk_tokens=chunk, sorted_dev=off).cpu()
        assert torch.equal(ident, torch.arange(len(lens) * max_splits, dtype=torch.int32))
        on = torch.ones(1, dtype=torch.int32, device=DEV)
        assert sorted(ops.decode_plan(sl, max_splits, chunk_tokens=chunk, sorted_dev=on).cpu()[: len(items)].tolist()) == sorted(items)


@pytest.mark.parametrize("Hq,Hkv,page_size", [(32, 8, 16), (8, 1, 64), (16, 8, 1)])
def test_decode_planned_launch_is_bit_identical(Hq, Hkv, page_size):
    """The plan only permutes work groups: a ragged batch gives the same bits with and without it, and matches
    the oracle; the split size comes from the host-side schedule model the backend uses for such batches."""
    ops = _ops()
    seq = [5000, 100, 900, 33, 2048, 1, 4097, 16, 17, 3000]
    case = build_case(Hq, Hkv, page_size, seq, seed=61 + Hq)
    chunk = ops.pick_split_planned(seq, 16, Hkv)
    assert chunk % 16 == 0 and -(-max(seq) // chunk) <= 16
    for max_splits, ch in [(16, chunk), (64, 128)]:
        base = run_decode(case, page_size, max_splits=max_splits, chunk_tokens=ch)
        check_decode(case, base)
        plan = ops.decode_plan(case["seq_lens"].to(DEV), max_splits, ch)
        assert torch.equal(run_decode(case, page_size, max_splits=max_splits, chunk_tokens=ch, plan=plan), base)


@pytest.mark.parametrize("Hq,Hkv,page_size", [(32, 8, 16), (4, 1, 64), (8, 2, 16), (16, 4, 32)])
def test_decode_four_heads_per_kv_head_on_the_matrix_core_kernel(Hq, Hkv, page_size, variants):
    """The matrix-core decode kernel is built for 4 query heads per kv head as well (a measurement override; by default
    only 8 heads per kv head take it): same results as the gather kernel to fp32 / bf16 rounding, and within the bound
    of the oracle, over splits, a launch plan and the cache-policy hint."""
    ops = _ops()
    seq = [1, 15, 16, 17, 100, 1023, 1024, 2500, 4097]
    case = build_case(Hq, Hkv, page_size, seq, seed=91 + Hq)
    plan = ops.decode_plan(case["seq_lens"].to(DEV), 16, 272)
    reuse = torch.tensor([0, 0, 16, 16, 64, 512, 0, 1024, 4096], dtype=torch.int32, device=DEV)
    for kw in (dict(), dict(max_splits=16, chunk_tokens=272), dict(max_splits=16, chunk_tokens=272, plan=plan, reuse_lens=reuse)):
        variants("decode_mfma", "0")
        gather = run_decode(case, page_size, **kw)
        variants("decode_mfma", "4")
        mfma = run_decode(case, page_size, **kw)
        check_decode(case, mfma)
        assert (mfma.float() - gather.float()).abs().max().item() <= 8e-3


@pytest.mark.parametrize("Hq,Hkv", [(64, 8), (32, 8), (8, 1), (4, 4)])
def test_decode_scores_that_grow_along_the_keys(Hq, Hkv):
    """Online-softmax bookkeeping of the decode kernels (gather kernel, matrix-core kernel for 8 heads per kv head,
    their split merge) under running maxima that jump by hundreds of units: K rows in bands of 512 positions scaled by
    1, 12, 1, 40, 90.  A maximum that misses part of the keys would leave probabilities far above 1 and overflow."""
    ps = 16
    seq = [300, 1200, 2500, 2560]
    case = build_case(Hq, Hkv, ps, seq, seed=123 + Hq)
    kv = case["kv_full"].float()
    for b, S in enumerate(seq):
        for band, f in enumerate([1.0, 12.0, 1.0, 40.0, 90.0]):
            lo, hi = band * 512, min((band + 1) * 512, S)
            if lo < hi:
                kv[case["r2t"][int(case["req_idx"][b]), lo:hi].long(), 0] *= f
    kv = kv.to(torch.bfloat16)
    case["kv_full"], case["k_cache"], case["v_cache"] = kv, kv[:, 0], kv[:, 1]
    for kw in (dict(), dict(max_splits=8, chunk_tokens=256)):
        check_decode(case, run_decode(case, ps, **kw))


@pytest.mark.parametrize("waves", ["4", "8"])
@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (16, 2)])
def test_decode_matrix_core_kernel_with_four_and_eight_waves(Hq, Hkv, waves, variants):
    """The matrix-core decode kernel runs a (request, split) with 8 waves, or with 4 for one or two kv heads per rank on
    launches of about one work group per compute unit; the override forces either on the same cases (ragged lengths,
    splits, a launch plan, page sizes 16 and 64)."""
    ops = _ops()
    variants("decode_mfma_waves", waves)
    for ps, seq in ((16, [1, 15, 16, 17, 100, 1023, 1024, 2500, 4097]), (64, [5000, 64, 65, 8191])):
        case = build_case(Hq, Hkv, ps, seq, seed=211 + Hq + ps)
        plan = ops.decode_plan(case["seq_lens"].to(DEV), 16, 272)
        for kw in (dict(), dict(max_splits=16, chunk_tokens=272), dict(max_splits=16, chunk_tokens=272, plan=plan)):
            check_decode(case, run_decode(case, ps, **kw))


def test_decode_long_context_split():
    case = build_case(8, 1, 64, [20000, 3], seed=9)
    check_decode(case, run_decode(case, 64, max_splits=128, chunk_tokens=256))


def test_decode_errors():
    ops = _ops()
    case = build_case(16, 8, 1, [5], seed=1)
    with pytest.raises(ValueError):
        ops.paged_decode(case["q"], case["k_cache"].to(DEV), case["v_cache"].to(DEV), case["r2t"].to(DEV),
                         case["req_idx"].to(DEV), case["seq_lens"].to(DEV), 1.0, 1)  # q on the CPU
    q96 = torch.zeros((1, 16, 96), dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros((8, 8, 96), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="head dim"):
        ops.paged_decode(q96, kc, kc.clone(), case["r2t"].to(DEV), case["req_idx"].to(DEV), case["seq_lens"].to(DEV), 1.0, 1)
    q3 = torch.zeros((1, 24, 128), dtype=torch.bfloat16, device=DEV)  # group of 3
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.paged_decode(q3, case["k_cache"].to(DEV).contiguous(), case["v_cache"].to(DEV).contiguous(), case["r2t"].to(DEV),
                         case["req_idx"].to(DEV), case["seq_lens"].to(DEV), 1.0, 1)


# ------------------------------------------------------------------------------------------------ extend
def build_extend_case(Hq, Hkv, page_size, prefix_lens, input_lens, seed):
    synth, workload = _synth()
    D = 128
    seq = [p + n for p, n in zip(prefix_lens, input_lens)]
    bs = len(seq)
    pool_pages = sum(-(-s // page_size) for s in seq) + 8
    pt = workload.make_page_table(seq, page_size, pool_pages, seed)
    slots = (pool_pages + 1) * page_size
    kv = synth.normal((slots, 2, Hkv, D), seed + 1)
    q = synth.normal((sum(input_lens), Hq, D), seed + 3)
    return dict(q=q, kv=kv, r2t=pt.req_to_token, req_idx=torch.arange(bs, dtype=torch.int32), seq_lens=pt.seq_lens,
                prefix=torch.tensor(prefix_lens, dtype=torch.int32), inp=torch.tensor(input_lens, dtype=torch.int32))


def run_extend(case, page_size, max_seqlen_k=0, planned=False, workspace=None):
    ops = _ops()
    kvd = case["kv"].to(DEV)
    cu = torch.nn.functional.pad(torch.cumsum(case["inp"], 0, dtype=torch.int32), (1, 0)).to(DEV)
    plan = ops.extend_plan(cu, case["prefix"].to(DEV), int(case["inp"].max())) if planned else None
    out = ops.paged_extend(case["q"].to(DEV), kvd[:, 0], kvd[:, 1], case["r2t"].to(DEV), case["req_idx"].to(DEV), cu,
                           case["prefix"].to(DEV), int(case["inp"].max()), 128 ** -0.5, page_size, max_seqlen_k=max_seqlen_k,
                           plan=plan, workspace=workspace)
    torch.cuda.synchronize()
    return out.cpu()


def check_extend(case, out):
    scale = 128 ** -0.5
    kc, vc = case["kv"][:, 0], case["kv"][:, 1]
    ref = orc.paged_extend_ref(case["q"], kc, vc, case["r2t"], case["req_idx"], case["seq_lens"], case["prefix"], case["inp"], scale)
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= 2e-2, f"vs oracle: max abs err {err}"
    start = 0
    for b in range(len(case["seq_lens"])):
        S, P, n = int(case["seq_lens"][b]), int(case["prefix"][b]), int(case["inp"][b])
        exact = orc.paged_attention_fp64(case["q"][start: start + n], kc, vc, case["r2t"][b, :S], scale,
                                         q_positions=torch.arange(P, P + n))
        # the extend kernel rounds the probabilities to bf16 for the P.V product (as the reference's SDPA does); on rows
        # with a handful of keys those roundings do not average out, hence twice the decode bound
        assert_close_fp64(out[start: start + n], exact, f"request {b} (prefix {P}, new {n})", slack=2.0)
        start += n


EXTEND_FORMS = ["one-head", "two-heads-four-waves", "two-heads-eight-waves"]


def _force_extend_form(variants, form):
    """The work-group shapes of the three-slot extend kernel: one query head per group (4 waves), two heads per group as
    4 waves that own both heads each (the default for an even number of heads per kv head), or as 8 waves that own one
    head each.  Pinned through msgl_set_variant; the fixture releases every choice after the test."""
    variants("extend_heads", "1" if form == "one-head" else "2")
    variants("extend_wide", "0" if form == "two-heads-eight-waves" else "1")


@pytest.mark.parametrize("form", EXTEND_FORMS)
@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (8, 1), (4, 4)])
@pytest.mark.parametrize("page_size", [1, 16, 64])
def test_extend(Hq, Hkv, page_size, form, variants):
    _force_extend_form(variants, form)
    ps = page_size
    prefix = [0, 0, ps * 2, 0, 64 * ps // ps * (ps if ps > 1 else 1), 0]
    prefix = [p - p % ps for p in prefix]
    inputs = [1, 31, 33, 128, 200, 300]
    case = build_extend_case(Hq, Hkv, ps, prefix, inputs, seed=3 + Hq + ps)
    check_extend(case, run_extend(case, ps))


@pytest.mark.parametrize("form", EXTEND_FORMS)
def test_extend_long_prefix(form, variants):
    _force_extend_form(variants, form)
    case = build_extend_case(16, 8, 16, [2048, 0], [130, 700], seed=21)
    check_extend(case, run_extend(case, 16))


@pytest.mark.parametrize("form", EXTEND_FORMS)
def test_extend_scores_that_grow_along_the_keys(form, variants):
    """Online-softmax bookkeeping under a running maximum that keeps moving: the keys come in bands of 256 positions
    whose rows are scaled by 1, 3, 1, 14, 1, 40, 80, so the row maxima jump by a few units (below the growth at which
    the four-wave kernel moves its reference maximum: probabilities above 1), by tens of units (every kernel rescales
    its output accumulators in the middle of its steady state), and again inside the rows' own causal range."""
    _force_extend_form(variants, form)
    ps = 16
    prefix, inputs = [1536, 256 + 32], [384, 300]
    case = build_extend_case(8, 2, ps, prefix, inputs, seed=77)
    kv = case["kv"].float()
    for b in range(2):
        for band, f in enumerate([1.0, 3.0, 1.0, 14.0, 1.0, 40.0, 80.0]):
            lo, hi = band * 256, min((band + 1) * 256, int(case["seq_lens"][b]))
            if lo < hi:
                kv[case["r2t"][b, lo:hi].long(), 0] *= f
    case["kv"] = kv.to(torch.bfloat16)
    check_extend(case, run_extend(case, ps))


@pytest.mark.parametrize("form", EXTEND_FORMS)
def test_extend_few_queries_on_long_unaligned_prefixes(form, variants):
    """Decode-like calls through the extend entry point (chunked prefill tails): 1-5 new tokens on prefixes that are
    not multiples of the 64-key tile, so almost every wave of the work group has nothing to do."""
    _force_extend_form(variants, form)
    case = build_extend_case(8, 2, 1, [4097, 33, 63, 64, 0], [1, 5, 2, 1, 1], seed=41)
    check_extend(case, run_extend(case, 1))


@pytest.mark.parametrize("Hq,Hkv,forced", [(8, 1, None), (8, 1, "3"), (2, 1, None), (4, 4, "2")])
def test_extend_key_split_matches_unsplit_and_oracle(Hq, Hkv, forced, variants):
    """Few work items over long key ranges: with a workspace the keys of every item are cut into runs, one work group
    each, and joined by a second kernel.  Compared with the oracle and with the unsplit launch (which sums the same
    products in another order: fp32 rounding only).  Covers runs that end before a row's first visible key, runs
    that are empty for the early query tiles, prefixes that are no multiple of a key tile, and ragged batches."""
    ops = _ops()
    if forced:
        variants("extend_key_split", forced)
    ps = 16
    prefix, inputs = [2048 + 16 * 3, 0, 4096], [200, 2300, 1]
    case = build_extend_case(Hq, Hkv, ps, prefix, inputs, seed=61 + Hq)
    plain = run_extend(case, ps, max_seqlen_k=max(p + n for p, n in zip(prefix, inputs)))
    check_extend(case, plain)
    T = sum(inputs)
    ws = torch.empty(ops.extend_workspace_bytes(T, Hq, 128), dtype=torch.uint8, device=DEV)
    ws.view(torch.float32).fill_(float("nan"))  # a partial that is read without having been written shows
    for planned in (False, True):
        split = run_extend(case, ps, max_seqlen_k=max(p + n for p, n in zip(prefix, inputs)), planned=planned, workspace=ws)
        check_extend(case, split)
        assert (split.float() - plain.float()).abs().max().item() <= 1e-2


@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (8, 1), (4, 4)])
@pytest.mark.parametrize("page_size", [1, 16])
def test_extend_short_requests_two_slot_kernel(Hq, Hkv, page_size, variants):
    """With the longest request announced (max_seqlen_k) and short, the library takes the kernel with a two-slot
    ring and one head per work group; 0 (unknown) takes a three-slot one.  Same oracle, same bar.  The two-slot and
    the eight-wave three-slot kernel agree to the last bit (same arithmetic in the same order, only the staging
    differs); the four-wave three-slot kernel (two heads per wave, the default for an even number of heads per kv
    head) lets a row's reference maximum lag behind its running maximum, so it agrees to rounding."""
    ps = page_size
    prefix = [0, 0, 2 * ps, 0, 16 * ps if ps > 1 else 64, 0, 448]
    inputs = [1, 31, 33, 128, 200, 300, 320]
    case = build_extend_case(Hq, Hkv, ps, prefix, inputs, seed=5 + Hq + ps)
    kmax = max(p + n for p, n in zip(prefix, inputs))
    assert kmax <= 768
    out = run_extend(case, ps, max_seqlen_k=kmax)
    check_extend(case, out)
    wide = run_extend(case, ps, max_seqlen_k=0)
    check_extend(case, wide)
    assert (wide.float() - out.float()).abs().max().item() <= 1e-2
    variants("extend_wide", "0")
    assert torch.equal(out, run_extend(case, ps, max_seqlen_k=0))


def test_extend_plan_orders_items_longest_first():
    """The device-built launch order against a numpy restatement: same items, non-increasing key-tile counts,
    -1 tail; ragged batch with empty-prefix, long-prefix and single-token requests, and a batch wider than the
    1024 threads of the planning kernel."""
    ops = _ops()
    for prefix, inp in [([0, 1024, 0, 4096, 0, 0, 2048, 0], [100, 300, 2000, 50, 700, 4000, 128, 1500]),
                        ([0], [1]), ([70000, 3], [129, 128]),
                        (list(range(0, 3000, 2)), [1 + (7 * i) % 300 for i in range(1500)])]:
        cu = torch.nn.functional.pad(torch.cumsum(torch.tensor(inp, dtype=torch.int32), 0, dtype=torch.int32), (1, 0))
        pre = torch.tensor(prefix, dtype=torch.int32)
        plan = ops.extend_plan(cu.to(DEV), pre.to(DEV), max(inp)).cpu()
        items, tiles = orc.extend_plan_ref(cu, pre, max(inp))
        n_real = len(items)
        assert plan.numel() == len(prefix) * (-(-max(inp) // 128))
        assert (plan[n_real:] == -1).all()
        got = plan[:n_real].tolist()
        assert sorted(got) == sorted(items)
        cost = dict(zip(items, tiles))
        got_cost = [min(cost[i], 2047) for i in got]
        assert got_cost == sorted(got_cost, reverse=True)


@pytest.mark.parametrize("forced_ring", [None, "2"])
def test_extend_planned_launch_is_bit_identical(forced_ring, variants):
    """The plan only permutes work groups: same bits with and without it, on a ragged batch with shared shapes."""
    if forced_ring:
        variants("extend_ring", forced_ring)
    case = build_extend_case(16, 8, 16, [0, 1024, 0, 2048, 0, 16], [100, 300, 900, 50, 1, 128], seed=51)
    base = run_extend(case, 16)
    check_extend(case, base)
    assert torch.equal(base, run_extend(case, 16, planned=True))


def test_extend_full_size_rows_match_decode_and_fp64():
    """A 4096-token prompt on top of a 1024-token cached prefix, Qwen3-8B heads (too large for the CPU oracle as
    a whole): sampled query rows must equal what the decode kernel returns for the same query over the same
    visible keys (row i of the extend == a decode step at length prefix + i + 1), and an fp64 evaluation."""
    ops = _ops()
    ps, P, n, Hq, Hkv = 16, 1024, 4096, 32, 8
    case = build_extend_case(Hq, Hkv, ps, [P], [n], seed=31)
    out = run_extend(case, ps)
    rows = [0, 1, 31, 32, 63, 64, 127, 128, 1000, 2047, 2048, 4000, 4095]
    kvd = case["kv"].to(DEV)
    q = case["q"][rows].contiguous().to(DEV)
    lens = torch.tensor([P + r + 1 for r in rows], dtype=torch.int32, device=DEV)
    idx = torch.zeros(len(rows), dtype=torch.int32, device=DEV)
    dec = ops.paged_decode(q, kvd[:, 0], kvd[:, 1], case["r2t"].to(DEV), idx, lens, 128 ** -0.5, ps, max_splits=8,
                           chunk_tokens=1024).cpu()
    assert (out[rows].float() - dec.float()).abs().max().item() <= 2e-2
    exact = orc.paged_attention_fp64(case["q"][rows], case["kv"][:, 0], case["kv"][:, 1], case["r2t"][0, : P + n], 128 ** -0.5,
                                     q_positions=torch.tensor([P + r for r in rows]))
    assert_close_fp64(out[rows], exact, "sampled rows of the 4096-token extend")


# ------------------------------------------------------------------------------------------------ layers
@pytest.mark.parametrize("cols", [128, 1024, 2048, 4096, 5120, 8192])
def test_rmsnorm(cols):
    ops = _ops()
    synth, _ = _synth()
    rows = 37
    x = synth.normal((rows, cols), 1, std=2.0)
    w = synth.normal((cols,), 2, std=0.5) + 1
    ref = orc.rmsnorm_ref(x, w, 1e-6)
    out = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6).cpu()
    d = ulp_diff_bf16(out, ref)
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02, (d.max(), (d > 0).float().mean())
    # strided rows (a 128-wide head slice of a wider buffer)
    if cols == 128:
        big = synth.normal((rows, 3 * cols), 3)
        xs = big[:, cols: 2 * cols]
        out = ops.rmsnorm(big.to(DEV)[:, cols: 2 * cols], w.to(DEV), 1e-6).cpu()
        d = ulp_diff_bf16(out, orc.rmsnorm_ref(xs, w, 1e-6))
        assert d.max() <= 1


@pytest.mark.parametrize("cols", [1024, 4096, 5120])
def test_fused_add_rmsnorm(cols):
    ops = _ops()
    synth, _ = _synth()
    rows = 19
    x, r = synth.normal((rows, cols), 4), synth.normal((rows, cols), 5, std=3.0)
    w = synth.normal((cols,), 6, std=0.5) + 1
    y_ref, r_ref = orc.rmsnorm_ref(x, w, 1e-6, residual=r)
    xd, rd = x.to(DEV), r.to(DEV)
    ops.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
    assert torch.equal(rd.cpu(), r_ref), "residual' = bf16(x + residual) must be bit exact"
    d = ulp_diff_bf16(xd.cpu(), y_ref)
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02


def test_silu_and_mul():
    ops = _ops()
    synth, _ = _synth()
    for rows, n in [(5, 3072), (64, 12288), (3, 8)]:
        x = synth.normal((rows, 2 * n), 7, std=2.0)
        ref = orc.silu_and_mul_ref(x)
        out = ops.silu_and_mul(x.to(DEV)).cpu()
        d = ulp_diff_bf16(out, ref)
        assert d.max() <= 1 and (d > 0).float().mean() < 0.01, (d.max(), (d > 0).float().mean())


def test_rope_bit_exact_and_fp32():
    ops = _ops()
    synth, _ = _synth()
    D, Hq, Hkv, T = 128, 16, 8, 50
    cache = orc.build_cos_sin_cache(D, 40960, 1e6)
    pos = torch.cat([torch.tensor([0, 1, 40959], dtype=torch.int32), synth.uniform_int(T - 3, 8, 0, 40960).to(torch.int32)])
    qkv = synth.normal((T, (Hq + 2 * Hkv) * D), 9)
    q, k = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D]
    q_ref, k_ref = orc.rope_ref(pos, q, k, cache, D)
    qkv_d = qkv.to(DEV)
    qd, kd = qkv_d[:, : Hq * D], qkv_d[:, Hq * D: (Hq + Hkv) * D]
    ops.rope_neox_(pos.to(DEV), qd, kd, cache.to(DEV), D, reference_rounding=True)
    assert torch.equal(qd.cpu(), q_ref) and torch.equal(kd.cpu(), k_ref), "reference-rounding RoPE must be bit exact"
    assert torch.equal(qkv_d[:, (Hq + Hkv) * D:].cpu(), qkv[:, (Hq + Hkv) * D:]), "v part must be untouched"
    qkv_d = qkv.to(DEV)
    qd, kd = qkv_d[:, : Hq * D], qkv_d[:, Hq * D: (Hq + Hkv) * D]
    ops.rope_neox_(pos.to(DEV), qd, kd, cache.to(DEV), D, reference_rounding=False)
    q32, k32 = orc.rope_fp32_ref(pos, q, k, cache, D)
    assert ulp_diff_bf16(qd.cpu(), q32).max() <= 1 and ulp_diff_bf16(kd.cpu(), k32).max() <= 1


def test_store_kv_bit_exact():
    ops = _ops()
    synth, _ = _synth()
    Hkv, D, T, slots = 8, 128, 40, 300
    qkv = synth.normal((T, (16 + 2 * Hkv) * D), 11)
    k = qkv[:, 16 * D: (16 + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (16 + Hkv) * D:].view(T, Hkv, D)
    loc = (synth.permutation(slots - 1, 12)[:T] + 1).to(torch.int32)
    for layout in ("interleaved", "split"):
        if layout == "interleaved":
            kv = torch.zeros((slots, 2, Hkv, D), dtype=torch.bfloat16)
            kc, vc = kv[:, 0], kv[:, 1]
            kvd = kv.to(DEV)
            kcd, vcd = kvd[:, 0], kvd[:, 1]
        else:
            kc, vc = torch.zeros((slots, Hkv, D), dtype=torch.bfloat16), torch.zeros((slots, Hkv, D), dtype=torch.bfloat16)
            kcd, vcd = kc.to(DEV), vc.to(DEV)
        orc.store_kv_ref(kc, vc, loc, k, v)
        qd = qkv.to(DEV)
        ops.store_kv(qd[:, 16 * D: (16 + Hkv) * D].view(T, Hkv, D), qd[:, (16 + Hkv) * D:].view(T, Hkv, D), loc.to(DEV), kcd, vcd)
        assert torch.equal(kcd.cpu(), kc) and torch.equal(vcd.cpu(), vc)


@pytest.mark.parametrize("T", [33, 20000])  # 20000 tokens take the rows-of-8-heads kernel
@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (8, 1)])
def test_fused_qknorm_rope_store(Hq, Hkv, T):
    ops = _ops()
    synth, _ = _synth()
    D, slots = 128, T + 167
    cache = orc.build_cos_sin_cache(D, 4096, 1e6)
    pos = synth.uniform_int(T, 13, 0, 4096).to(torch.int32)
    qkv = synth.normal((T, (Hq + 2 * Hkv) * D), 14, std=1.5)
    qw, kw = synth.normal((D,), 15, std=0.3) + 1, synth.normal((D,), 16, std=0.3) + 1
    loc = (synth.permutation(slots - 1, 17)[:T] + 1).to(torch.int32)
    # oracle: unfused op sequence of the reference (qwen3.py:88-94 then set_kv_buffer)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    qn = orc.rmsnorm_ref(q.reshape(-1, D), qw, 1e-6).view(q.shape)
    kn = orc.rmsnorm_ref(k.reshape(-1, D), kw, 1e-6).view(k.shape)
    q_ref, k_ref = orc.rope_ref(pos, qn, kn, cache, D)
    kc, vc = torch.zeros((slots, Hkv, D), dtype=torch.bfloat16), torch.zeros((slots, Hkv, D), dtype=torch.bfloat16)
    orc.store_kv_ref(kc, vc, loc, k_ref, v)
    kvd = torch.zeros((slots, 2, Hkv, D), dtype=torch.bfloat16, device=DEV)
    qkv_d = qkv.to(DEV)
    ops.qknorm_rope_store_(qkv_d, qw.to(DEV), kw.to(DEV), 1e-6, pos.to(DEV), cache.to(DEV), Hq, Hkv, D, loc=loc.to(DEV),
                           k_cache=kvd[:, 0], v_cache=kvd[:, 1])
    got = qkv_d.cpu()
    dq = ulp_diff_bf16(got[:, : Hq * D], q_ref)
    dk = ulp_diff_bf16(got[:, Hq * D: (Hq + Hkv) * D], k_ref)
    # the norm may differ by one bf16 ulp on few elements (sum order / rsqrt); rope then mixes pairs, so allow 2 —
    # in ulps, or, where the rotation cancels to a small number, the same distance at the scale of its operands
    far_q = (dq > 2) & ((got[:, : Hq * D].float() - q_ref.float()).abs() > 0.04)
    far_k = (dk > 2) & ((got[:, Hq * D: (Hq + Hkv) * D].float() - k_ref.float()).abs() > 0.04)
    assert not far_q.any() and not far_k.any() and (dq > 0).float().mean() < 0.03, (dq.max(), dk.max(), (dq > 0).float().mean())
    if T > 1000:
        # the rows-of-8-heads kernel and the head-per-row kernel (taken for few tokens) do the same arithmetic: bit equal
        kv2 = torch.zeros_like(kvd)
        parts = []
        for a in range(0, T, 256):
            part = qkv[a: a + 256].to(DEV)
            ops.qknorm_rope_store_(part, qw.to(DEV), kw.to(DEV), 1e-6, pos[a: a + 256].to(DEV), cache.to(DEV), Hq, Hkv, D,
                                   loc=loc[a: a + 256].to(DEV), k_cache=kv2[:, 0], v_cache=kv2[:, 1])
            parts.append(part)
        assert torch.equal(torch.cat(parts), qkv_d) and torch.equal(kv2, kvd)
    assert torch.equal(got[:, (Hq + Hkv) * D:], qkv[:, (Hq + Hkv) * D:])
    assert torch.equal(kvd[:, 1].cpu(), vc), "v rows must be copied bit exact"
    assert torch.equal(kvd[:, 0].cpu()[loc.long()].reshape(T, -1), got[:, Hq * D: (Hq + Hkv) * D]), "stored k = rotated k"


def test_argmax():
    ops = _ops()
    synth, _ = _synth()
    logits = synth.normal((7, 151936), 18)
    logits[2, 100] = logits[2, 5000] = 9.0  # tie: first index wins
    logits[3, 77] = float("nan")
    for t in (logits, logits.float()):
        got = ops.argmax(t.to(DEV)).cpu()
        assert torch.equal(got, orc.argmax_ref(t)), (got, orc.argmax_ref(t))
        idx, val = ops.argmax(t.to(DEV), return_values=True)
        want = t.float().gather(1, orc.argmax_ref(t).long()[:, None])[:, 0]
        assert torch.equal(idx.cpu(), got)
        assert torch.equal(val.cpu().nan_to_num(nan=123.0), want.nan_to_num(nan=123.0)), "the winning logit itself, bit for bit"


# The layer kernels take other forms at prefill sizes (a wave per row, several rows / heads / tokens per lane group,
# non-temporal accesses for tensors far beyond the caches); same oracle, same bounds.  Odd row counts exercise the tails.
@pytest.mark.parametrize("cols", [1024, 2048, 2560, 4096, 5120, 8192])
def test_rmsnorm_many_rows(cols):
    ops = _ops()
    synth, _ = _synth()
    rows = 4099
    x, r = synth.normal((rows, cols), 41, std=2.0), synth.normal((rows, cols), 42, std=3.0)
    w = synth.normal((cols,), 43, std=0.5) + 1
    d = ulp_diff_bf16(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6).cpu(), orc.rmsnorm_ref(x, w, 1e-6))
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02, (d.max(), (d > 0).float().mean())
    y_ref, r_ref = orc.rmsnorm_ref(x, w, 1e-6, residual=r)
    xd, rd = x.to(DEV), r.to(DEV)
    ops.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
    assert torch.equal(rd.cpu(), r_ref), "residual' = bf16(x + residual) must be bit exact"
    d = ulp_diff_bf16(xd.cpu(), y_ref)
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02


def test_rmsnorm_streamed_sizes():
    """[16384 + 3, 4096] plain (in + out >= 512 MiB: non-temporal form) and fused-add, and 128-wide head rows beyond
    the count at which a lane group takes four rows, strided as the q part of a qkv buffer."""
    ops = _ops()
    synth, _ = _synth()
    rows, cols = 16387, 4096
    x, r = synth.normal((rows, cols), 44, std=2.0, device=DEV), synth.normal((rows, cols), 45, std=3.0, device=DEV)
    w = synth.normal((cols,), 46, std=0.5) + 1
    xh, rh = x.cpu(), r.cpu()
    d = ulp_diff_bf16(ops.rmsnorm(x, w.to(DEV), 1e-6).cpu(), orc.rmsnorm_ref(xh, w, 1e-6))
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02
    y_ref, r_ref = orc.rmsnorm_ref(xh, w, 1e-6, residual=rh)
    ops.fused_add_rmsnorm(x, r, w.to(DEV), 1e-6)
    assert torch.equal(r.cpu(), r_ref)
    d = ulp_diff_bf16(x.cpu(), y_ref)
    assert d.max() <= 1 and (d > 0).float().mean() < 0.02
    wd = synth.normal((128,), 47, std=0.3) + 1
    for rows, stride in ((163961, 128), (140003, 256), (1048579, 128)):  # dense; 128-wide slices of wider rows; streamed
        buf = synth.normal((rows, stride), 48, std=1.5, device=DEV)
        got = ops.rmsnorm(buf[:, :128], wd.to(DEV), 1e-6).cpu()
        d = ulp_diff_bf16(got, orc.rmsnorm_ref(buf.cpu()[:, :128], wd, 1e-6))
        assert d.max() <= 1 and (d > 0).float().mean() < 0.02


@pytest.mark.parametrize("rows,n", [(4099, 12288), (8193, 12288), (1367, 3072)])
def test_silu_and_mul_many_rows(rows, n):
    ops = _ops()
    synth, _ = _synth()
    x = synth.normal((rows, 2 * n), 49, std=2.0, device=DEV)
    d = ulp_diff_bf16(ops.silu_and_mul(x).cpu(), orc.silu_and_mul_ref(x.cpu()))
    assert d.max() <= 1 and (d > 0).float().mean() < 0.01, (d.max(), (d > 0).float().mean())


@pytest.mark.parametrize("T", [7001, 26301])  # runs of four heads per lane group; the larger one streamed
def test_rope_many_tokens_bit_exact(T):
    ops = _ops()
    synth, _ = _synth()
    D, Hq, Hkv = 128, 32, 8
    cache = orc.build_cos_sin_cache(D, 40960, 1e6)
    pos = synth.uniform_int(T, 50, 0, 40960).to(torch.int32)
    qkv_d = synth.normal((T, (Hq + 2 * Hkv) * D), 51, device=DEV)
    qkv = qkv_d.cpu()
    q_ref, k_ref = orc.rope_ref(pos, qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], cache, D)
    for ref_rounding in (True, False):
        work = qkv_d.clone()
        qd, kd = work[:, : Hq * D], work[:, Hq * D: (Hq + Hkv) * D]
        ops.rope_neox_(pos.to(DEV), qd, kd, cache.to(DEV), D, reference_rounding=ref_rounding)
        if ref_rounding:
            assert torch.equal(qd.cpu(), q_ref) and torch.equal(kd.cpu(), k_ref), "reference-rounding RoPE must be bit exact"
        else:
            q32, k32 = orc.rope_fp32_ref(pos, qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], cache, D)
            assert ulp_diff_bf16(qd.cpu(), q32).max() <= 1 and ulp_diff_bf16(kd.cpu(), k32).max() <= 1
        assert torch.equal(work[:, (Hq + Hkv) * D:], qkv_d[:, (Hq + Hkv) * D:]), "v part must be untouched"
    # 5 heads per token (not a multiple of the run length): the last run of every token is partial
    small = synth.normal((T, 5 * D), 52, device=DEV)
    sh = small.cpu()
    q_ref, k_ref = orc.rope_ref(pos, sh[:, : 4 * D], sh[:, 4 * D:], cache, D)
    ops.rope_neox_(pos.to(DEV), small[:, : 4 * D], small[:, 4 * D:], cache.to(DEV), D)
    assert torch.equal(small[:, : 4 * D].cpu(), q_ref) and torch.equal(small[:, 4 * D:].cpu(), k_ref)


def test_store_kv_many_tokens_bit_exact():
    ops = _ops()
    synth, _ = _synth()
    for Hkv, T in ((8, 4101), (1, 33001)):
        D, slots = 128, T + 300
        qkv = synth.normal((T, (16 + 2 * Hkv) * D), 53, device=DEV)
        loc = (synth.permutation(slots - 1, 54)[:T] + 1).to(torch.int32)
        kvd = torch.zeros((slots, 2, Hkv, D), dtype=torch.bfloat16, device=DEV)
        ops.store_kv(qkv[:, 16 * D: (16 + Hkv) * D].view(T, Hkv, D), qkv[:, (16 + Hkv) * D:].view(T, Hkv, D), loc.to(DEV), kvd[:, 0], kvd[:, 1])
        kc, vc = torch.zeros((slots, Hkv, D), dtype=torch.bfloat16), torch.zeros((slots, Hkv, D), dtype=torch.bfloat16)
        h = qkv.cpu()
        orc.store_kv_ref(kc, vc, loc, h[:, 16 * D: (16 + Hkv) * D], h[:, (16 + Hkv) * D:])
        assert torch.equal(kvd[:, 0].cpu(), kc) and torch.equal(kvd[:, 1].cpu(), vc)


# ------------------------------------------------------------------------------------------------ batch preparation
def _prepare_case(bs, ps, seed, max_ctx=512):
    synth, _ = _synth()
    lens = (synth.uniform_int(bs, seed, 1, max_ctx - 40)).tolist()
    if bs >= 4:
        lens[0], lens[1], lens[2], lens[3] = ps, ps - 1 if ps > 1 else 1, 2 * ps, 1  # page boundaries in the first steps
    rows = synth.permutation(bs + 5, seed + 1)[:bs].tolist()                        # table rows in no particular order
    total_pages = bs * (max_ctx // ps + 2)
    free_pages = (synth.permutation(total_pages, seed + 2) + 1).tolist()            # a free list in allocator order
    last = (synth.uniform_int(bs, seed + 3, ps, ps * total_pages)).tolist()
    r2t = torch.zeros((bs + 5, max_ctx), dtype=torch.int32)
    return lens, rows, free_pages, last, r2t


@pytest.mark.parametrize("bs,ps", [(1, 1), (3, 16), (300, 1), (300, 16), (1500, 16), (2500, 64)])
def test_prepare_decode_matches_oracle(bs, ps):
    ops = _ops()
    lens, rows, free_pages, last, r2t = _prepare_case(bs, ps, 11 + bs + ps)
    d = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
    seq_d, rows_d, free_d, last_d, r2t_d = d(lens), d(rows), d(free_pages), d(last), r2t.to(DEV)
    head_d, status = d([7, len(free_pages)]), d([0])
    pos_d, loc_d = torch.empty(bs, dtype=torch.int32, device=DEV), torch.empty(bs, dtype=torch.int32, device=DEV)
    toks = d(list(range(100, 100 + bs)))
    ids_d = torch.empty(bs, dtype=torch.int32, device=DEV)
    head = 7
    # the page half of the table (one entry per page), kept by the kernel when the caller has one
    r2p = torch.zeros((r2t.shape[0], -(-r2t.shape[1] // ps)), dtype=torch.int32) if ps > 1 else None
    r2p_d = r2p.to(DEV) if r2p is not None else None
    for step in range(20):  # 20 steps: every request with page_size 16 crosses a page boundary at least once
        positions, loc, head = orc.prepare_decode_ref(rows, lens, last, free_pages, head, r2t, ps, req_to_page=r2p)
        ops.prepare_decode(rows_d, seq_d, last_d, free_d, head_d, r2t_d, pos_d, loc_d, status, ps, next_tokens=toks, input_ids=ids_d,
                           req_to_page=r2p_d)
        assert pos_d.tolist() == positions and loc_d.tolist() == loc, f"step {step}"
    assert status.item() == 0 and head_d.tolist() == [head, len(free_pages)]
    assert seq_d.tolist() == lens and last_d.tolist() == last
    assert torch.equal(r2t_d.cpu(), r2t)
    if r2p is not None:
        assert torch.equal(r2p_d.cpu(), r2p) and int((r2p != 0).sum()) > 0
    assert torch.equal(ids_d, toks)


def test_prepare_decode_refuses_as_a_whole():
    ops = _ops()
    d = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
    # three requests, two of them need a page, one page left: refused; lengths, table and free list untouched, and
    # the step's K/V writes pointed at the dummy slot
    seq, rows, last = d([16, 5, 32]), d([0, 1, 2]), d([31, 100, 63])
    free, head, status = d([9, 4]), d([1, 2]), d([0])
    r2t = torch.full((3, 64), -1, dtype=torch.int32, device=DEV)
    pos, loc = d([-1, -1, -1]), d([-1, -1, -1])
    ops.prepare_decode(rows, seq, last, free, head, r2t, pos, loc, status, 16)
    assert status.item() == 1 and head.tolist() == [1, 2]
    assert seq.tolist() == [16, 5, 32] and pos.tolist() == [-1] * 3 and loc.tolist() == [0] * 3 and (r2t == -1).all()
    assert last.tolist() == [31, 100, 63]
    # sticky: a step the free list could serve is refused as well until the host has cleared the status word
    ops.prepare_decode(rows, d([3, 5, 7]), last, free, head, r2t, pos, loc, status, 16)
    assert status.item() == 1 and head.tolist() == [1, 2] and (r2t == -1).all()
    with pytest.raises(RuntimeError):
        orc.prepare_decode_ref([0, 1, 2], [16, 5, 32], [31, 100, 63], [9, 4], 1, torch.zeros((3, 64), dtype=torch.int32), 16)
    # a request that already fills its table row: refused with status 2
    status.zero_()
    seq2 = d([64, 5, 3])
    ops.prepare_decode(rows, seq2, last, d([9, 4, 5]), d([0, 3]), r2t, pos, loc, status, 16)
    assert status.item() == 2 and seq2.tolist() == [64, 5, 3] and (r2t == -1).all()
