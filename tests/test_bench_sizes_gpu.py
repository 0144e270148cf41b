"""Parity at the sizes bench.py times.  The secondary legs of the benchmark line (128k-token decode of a Qwen3-32B TP-8
shard, prefill on a 64k-token prefix with the key split, the 256-request mixed 1k-32k batch with radix-shared prefixes)
are built here by the very functions bench.py calls (build_decode_config / build_extend_config / mixed_radix_batch), so
what is checked is what is timed: same page tables, split sizes, launch plans, workspaces and kernel choices.
Bounds as everywhere: 2e-2 abs against the oracle (torch_attn_backend.py:28-182 restated) and one bf16 ulp + 4e-3
against an fp64 evaluation."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as orc  # noqa: E402
from test_kernels_gpu import assert_close_fp64  # noqa: E402

SCALE = 128 ** -0.5


def _bench():
    import bench
    return bench


def _request_kv(cfg, layer, b):
    """K and V rows of request b in position order, on the host: [S, Hkv, D] each."""
    S = int(cfg["seq_lens"][b])
    slots = cfg["req_to_token"][b, :S].long()
    kv = cfg["pools"][layer][slots].cpu()
    return kv[:, 0].contiguous(), kv[:, 1].contiguous()


def _fp64_decode_row(cfg, layer, b):
    k, v = _request_kv(cfg, layer, b)
    return orc.paged_attention_fp64(cfg["q"][b: b + 1].cpu(), k, v, torch.arange(k.shape[0]), SCALE)[0]


@pytest.mark.parametrize("bs", [1, 8])
def test_decode_131072_token_contexts_with_the_benchmarked_split(bs):
    """Qwen3-32B TP-8 shard (8 q / 1 kv head, page_size 64) at 131072-token contexts, batch 1 and 8: the split size and
    count ops.pick_split gives the benchmark (512 tokens x 256 splits at batch 1, 4096 x 32 at batch 8), the matrix-core
    kernel's four-wave groups and the many-split merge, against the oracle's per-request gather + SDPA on the whole
    batch and fp64 on the first and last request."""
    bench = _bench()
    cfg = bench.build_decode_config(8, 1, 64, [131072] * bs, layers=1)
    assert (cfg["chunk"], cfg["max_splits"]) == ((512, 256) if bs == 1 else (4096, 32))
    cfg["launch"](0)
    torch.cuda.synchronize()
    out = cfg["out"].cpu()
    kv = cfg["pools"][0].cpu()
    ref = orc.paged_decode_ref(cfg["q"].cpu(), kv[:, 0], kv[:, 1], cfg["req_to_token"].cpu(), cfg["req_idx"].cpu(), cfg["seq_lens"].cpu(), SCALE)
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= 2e-2, f"vs oracle: max abs err {err}"
    for b in sorted({0, bs - 1}):
        assert_close_fp64(out[b], _fp64_decode_row(cfg, 0, b), f"request {b} of {bs} at 131072 tokens")


def test_extend_4096_tokens_on_a_65536_token_prefix_with_the_key_split():
    """The long-context prefill leg: 4096 new tokens on a 65536-token cached prefix, 8 q / 1 kv head, page_size 64, with
    the key-split workspace the backend passes (NaN-filled first: a partial read before it is written shows).  Too large
    for the CPU oracle as a whole (S x S scores), so sampled query rows are compared with fp64 and with the decode
    kernel on the same visible keys (row i of the extend == a decode step at length prefix + i + 1)."""
    from mini_sglang_amd import ops

    bench = _bench()
    P, n, Hq, Hkv, ps = 65536, 4096, 8, 1, 64
    cfg = bench.build_extend_config(Hq, Hkv, ps, n, P)
    assert cfg["workspace"] is not None
    cfg["workspace"].view(torch.float32).fill_(float("nan"))
    cfg["launch"]()
    torch.cuda.synchronize()
    out = cfg["out"].cpu()
    assert torch.isfinite(out.float()).all()
    rows = [0, 1, 31, 32, 63, 64, 127, 128, 1000, 2047, 2048, 4000, 4095]
    kv = cfg["kv"]
    q = cfg["q"][rows].contiguous()
    lens = torch.tensor([P + r + 1 for r in rows], dtype=torch.int32, device="cuda")
    idx = torch.zeros(len(rows), dtype=torch.int32, device="cuda")
    dec = ops.paged_decode(q, kv[:, 0], kv[:, 1], cfg["req_to_token"], idx, lens, SCALE, ps, max_splits=32, chunk_tokens=4096).cpu()
    assert (out[rows].float() - dec.float()).abs().max().item() <= 2e-2
    slots = cfg["req_to_token"][0, : P + n].long()
    kvh = kv[slots].cpu()
    exact = orc.paged_attention_fp64(cfg["q"][rows].cpu(), kvh[:, 0].contiguous(), kvh[:, 1].contiguous(), torch.arange(P + n), SCALE,
                                     q_positions=torch.tensor([P + r for r in rows]))
    assert_close_fp64(out[rows], exact, "sampled rows of the 4096-token extend on a 65536-token prefix")


def test_mixed_radix_shared_batch_as_benchmarked():
    """Qwen3-8B heads, 256 requests with contexts log-uniform in 1k-32k, 16 groups of 16 on shared page-aligned 1-4k
    prefixes, page_size 16, exactly as bench.py builds the two legs (per-request reads with the launch plan; every
    group's prefix attended once, then the private tails): the two legs agree to 2e-2, three sampled requests (shortest,
    longest, one from the middle) match fp64 in both, and the per-request leg is invariant under a permutation of the
    batch (each request's output depends on the request alone, whatever the plan made of the launch order)."""
    from mini_sglang_amd import ops

    bench = _bench()
    lens, groups, prefix = bench.mixed_radix_batch(256)
    torch.manual_seed(1234)
    per_req = bench.build_decode_config(32, 8, 16, lens, groups, prefix, layers=1)
    per_req["launch"](0)
    torch.cuda.synchronize()
    out_a = per_req["out"].clone()
    torch.manual_seed(1234)  # same q and pool contents
    once = bench.build_decode_config(32, 8, 16, lens, groups, prefix, layers=1, share_once=True)
    assert torch.equal(once["q"], per_req["q"]) and torch.equal(once["req_to_token"], per_req["req_to_token"])
    once["launch"](0)
    torch.cuda.synchronize()
    out_b = once["out"].clone()
    assert per_req["plan"] is not None and once["cascade"] is not None
    assert (out_a.float() - out_b.float()).abs().max().item() <= 2e-2
    order = sorted(range(256), key=lambda i: lens[i])
    for b in (order[0], order[128], order[-1]):
        exact = _fp64_decode_row(per_req, 0, b)
        assert_close_fp64(out_a[b].cpu(), exact, f"request {b} ({lens[b]} tokens), per-request reads")
        assert_close_fp64(out_b[b].cpu(), exact, f"request {b} ({lens[b]} tokens), prefix once", slack=2.0)
    # permutation: request i of the permuted batch is request perm[i] of the original one
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(256, generator=g)
    pd = perm.to("cuda")
    c = per_req
    sl = c["seq_lens"][pd].contiguous()
    plan = ops.decode_plan(sl, c["max_splits"], c["chunk"])
    kvp = c["pools"][0]
    out_p = ops.paged_decode(c["q"][pd].contiguous(), kvp[:, 0], kvp[:, 1], c["req_to_token"], pd.to(torch.int32), sl, SCALE, 16,
                             workspace=c["workspace"], max_splits=c["max_splits"], chunk_tokens=c["chunk"],
                             reuse_lens=c["reuse_lens"][pd].contiguous(), plan=plan)
    torch.cuda.synchronize()
    assert torch.equal(out_p, out_a[pd])
