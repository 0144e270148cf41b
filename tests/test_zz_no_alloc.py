"""msgl_paged_decode_static (the decode entry point that allocates nothing: its split workspace is a static device
array inside the library) against msgl_paged_decode with a caller workspace, through the C ABI: bit-identical
outputs, on two calls in a row of which the second is smaller (the shared array keeps the first call's partials),
no device memory allocated by the calls, and a clear refusal beyond the array's size."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
DEV = "cuda"
D = 128


def _case(Hq, Hkv, ps, lens, seed):
    from mini_sglang_amd.runtime import synth, workload

    pages = sum(-(-n // ps) for n in lens) + 8
    pt = workload.make_page_table(lens, ps, pages, seed)
    kv = synth.normal(((pages + 1) * ps, 2, Hkv, D), seed + 1).to(DEV)
    q = synth.normal((len(lens), Hq, D), seed + 2).to(DEV)
    return q, kv[:, 0], kv[:, 1], pt.req_to_token.to(DEV), torch.arange(len(lens), dtype=torch.int32, device=DEV), pt.seq_lens.to(DEV)


def test_static_workspace_entry_point_matches_the_workspace_one():
    from mini_sglang_amd import ops
    from oracle import torch_oracle as orc

    scale = D ** -0.5
    # first call: 24 requests in up to 16 splits; second call: fewer, shorter requests and fewer splits
    calls = [(32, 8, 16, [3000, 17, 1, 2048, 999, 4000] * 4, 16, 256, 5), (32, 8, 16, [700, 33, 5], 4, 256, 9),
             (8, 1, 64, [5000, 64], 64, 128, 11), (16, 8, 1, [300, 1, 77], 1, 0, 13)]
    for Hq, Hkv, ps, lens, max_splits, chunk, seed in calls:
        q, kc, vc, r2t, idx, sl = _case(Hq, Hkv, ps, lens, seed)
        want = ops.paged_decode(q, kc, vc, r2t, idx, sl, scale, ps, max_splits=max_splits, chunk_tokens=chunk)
        out = torch.full_like(q, float("nan"))
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        ops.paged_decode_static(q, kc, vc, r2t, idx, sl, scale, ps, out, max_splits=max_splits, chunk_tokens=chunk)
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        assert torch.equal(out, want), f"{lens}: differs from msgl_paged_decode"
        if seed != 5:  # (the very first call on a device looks the array's address up; nothing is allocated after it)
            assert after[0] == before[0] and after[1] >= before[1], "the call must not allocate device memory"
        ref = orc.paged_decode_ref(q.cpu(), kc.cpu(), vc.cpu(), r2t.cpu(), idx.cpu(), sl.cpu(), scale)
        assert (out.cpu().float() - ref.float()).abs().max().item() <= 2e-2


def test_static_workspace_entry_point_under_graph_capture_and_its_limit():
    from mini_sglang_amd import ops

    scale = D ** -0.5
    q, kc, vc, r2t, idx, sl = _case(16, 8, 16, [1500, 40, 800, 16], 21)
    want = ops.paged_decode(q, kc, vc, r2t, idx, sl, scale, 16, max_splits=8, chunk_tokens=256)
    out = torch.zeros_like(q)
    ops.paged_decode_static(q, kc, vc, r2t, idx, sl, scale, 16, out, max_splits=8, chunk_tokens=256)  # warm-up outside capture
    torch.cuda.synchronize()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    # (inference mode, like every other capture in this suite: torch keeps per-process capture state that an earlier
    # capture under inference mode has turned into inference tensors)
    with torch.inference_mode():
        with torch.cuda.graph(g):
            ops.paged_decode_static(q, kc, vc, r2t, idx, sl, scale, 16, out, max_splits=8, chunk_tokens=256)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # 256 requests x 64 splits x 32 heads would need 272 MB of partials: refused, nothing launched
    need = ops.decode_workspace_bytes(256, 32, D, 64)
    assert need > ops.STATIC_WORKSPACE_BYTES
    qb = torch.zeros((256, 32, D), dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros((64, 2, 8, D), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="built-in one holds"):
        ops.paged_decode_static(qb, kb[:, 0], kb[:, 1], torch.zeros((256, 64), dtype=torch.int32, device=DEV),
                                torch.arange(256, dtype=torch.int32, device=DEV), torch.ones(256, dtype=torch.int32, device=DEV),
                                scale, 16, torch.empty_like(qb), max_splits=64, chunk_tokens=16)
