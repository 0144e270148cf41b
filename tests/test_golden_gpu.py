"""HIP path against the committed golden fixtures (outputs of the reference's torch-native path, see
oracle/gen_golden.py).  Tolerances: 2e-2 abs on attention outputs and on bf16 logits (the stated parity
bound), at most one bf16 ulp on norm / activation outputs, bit-exact for RoPE (reference rounding), for the
fused-add residual and for every integer output (slot assignments, page tables, token streams)."""
import pytest
import torch

import golden_cases as gc
from oracle import gen_golden as gg
from test_kernels_gpu import ulp_diff_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mini_sglang_amd import ops
    return ops


def test_layer_ops_vs_reference_fixture():
    ops = _ops()
    fx = gc.load("ops.pt")
    for rows, cols, seed in gg.OPS_CASES["rmsnorm"]:
        x, w = gc.rmsnorm_inputs(rows, cols, seed)
        d = ulp_diff_bf16(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6), fx[f"rmsnorm/{rows}x{cols}/{seed}"])
        assert d.max() <= 1, (rows, cols, d.max())
    for rows, cols, seed in gg.OPS_CASES["fused_add_rmsnorm"]:
        x, r, w = gc.fused_inputs(rows, cols, seed)
        xd, rd = x.to(DEV), r.to(DEV)
        ops.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
        assert torch.equal(rd.cpu(), fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/residual"])
        assert ulp_diff_bf16(xd, fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/y"]).max() <= 1
    for rows, n, seed in gg.OPS_CASES["silu_and_mul"]:
        d = ulp_diff_bf16(ops.silu_and_mul(gc.silu_inputs(rows, n, seed).to(DEV)), fx[f"silu_and_mul/{rows}x{n}/{seed}"])
        assert d.max() <= 1
    from mini_sglang_amd.layers.rotary_embedding import build_cos_sin_cache
    for T, Hq, Hkv, D, seed in gg.OPS_CASES["rope"]:
        pos, q, k = gc.rope_inputs(T, Hq, Hkv, D, seed)
        cache = build_cos_sin_cache(D, 40960, 1e6)
        key = f"rope/{T}x{Hq}x{Hkv}x{D}/{seed}"
        assert torch.equal(cache[[0, 1, 40959]], fx[key + "/cache_rows"])
        qd, kd = q.to(DEV), k.to(DEV)
        ops.rope_neox_(pos.to(DEV), qd, kd, cache.to(DEV), D, reference_rounding=True)
        assert torch.equal(qd.cpu(), fx[key + "/q"]) and torch.equal(kd.cpu(), fx[key + "/k"])


def test_attention_vs_reference_fixture():
    ops = _ops()
    fx = gc.load("attention.pt")
    scale = 128 ** -0.5
    for ci, case in enumerate(gg.ATTN_DECODE_CASES):
        c = gc.decode_inputs(case)
        kc, vc = c["kc"].to(DEV), c["vc"].to(DEV)
        ops.store_kv(c["k_new"].to(DEV), c["v_new"].to(DEV), c["loc"].to(DEV), kc, vc)
        for max_splits, chunk in [(1, 0), (8, 128)]:
            out = ops.paged_decode(c["q"].to(DEV).view(-1, c["Hq"], 128), kc, vc, c["pt"].req_to_token.to(DEV), c["req_idx"].to(DEV),
                                   c["pt"].seq_lens.to(DEV), scale, c["ps"], max_splits=max_splits, chunk_tokens=chunk)
            err = (out.cpu().float().reshape(len(case[3]), -1) - fx[f"decode/{ci}"].float()).abs().max().item()
            assert err <= 2e-2, f"decode case {ci} splits {max_splits}: {err}"
    for ci, case in enumerate(gg.ATTN_EXTEND_CASES):
        c = gc.extend_inputs(case)
        kc, vc = c["kc"].to(DEV), c["vc"].to(DEV)
        ops.store_kv(c["k_new"].to(DEV), c["v_new"].to(DEV), c["loc"].to(DEV), kc, vc)
        cu = torch.nn.functional.pad(torch.cumsum(c["inp"], 0, dtype=torch.int32), (1, 0)).to(DEV)
        out = ops.paged_extend(c["q"].to(DEV).view(-1, c["Hq"], 128), kc, vc, c["pt"].req_to_token.to(DEV), c["req_idx"].to(DEV), cu,
                               c["prefix"].to(DEV), int(c["inp"].max()), scale, c["ps"])
        got = out.cpu().float().reshape(out.shape[0], -1)[:, ::gg.EXTEND_COL_STRIDE]
        err = (got - fx[f"extend/{ci}"].float()).abs().max().item()
        assert err <= 2e-2, f"extend case {ci}: {err}"


def _check_e2e(got, fx, need_identical_stream, wide_tol=2e-2):
    # integer outputs: slot assignment and page tables never depend on the sampled tokens
    for key in ("out_cache_loc_first", "out_cache_loc_decode", "req_to_token_rows"):
        assert torch.equal(got[key], fx[key]), key
    assert got["req_to_page_matches_tokens"]
    same = (got["tokens"] == fx["tokens"])
    if need_identical_stream:
        assert bool(same.all()), f"greedy streams differ at {(~same).nonzero()[:4].tolist()}"
        steps = fx["tokens"].shape[1]
    else:
        # random head: compare up to the first step where the reference's own top-2 margin is below 4 bf16 ulps
        weak = (fx["margins"] < 0.125).any(dim=1).nonzero()
        steps = int(weak[0]) if weak.numel() else fx["tokens"].shape[1]
        assert bool(same[:, :steps].all())
        steps += 1  # the logits of the first weak step were still computed from identical inputs
    d_top = (got["top1_logits"][:steps] - fx["top1_logits"][:steps]).abs().max().item()
    d_sl = (got["logits_slices"][:steps].float() - fx["logits_slices"][:steps].float()).abs().max().item()
    # 2048 columns spread over the whole vocabulary, every step
    d_wide = (got["logits_wide"][:steps].float() - fx["logits_wide"][:steps].float()).abs().max().item()
    assert d_top <= 2e-2 and d_sl <= 2e-2 and d_wide <= wide_tol, (d_top, d_sl, d_wide)
    beyond_one_ulp = ((got["logits_wide"][:steps].float() - fx["logits_wide"][:steps].float()).abs() > 0.0157).float().mean().item()
    assert beyond_one_ulp <= 0.01, beyond_one_ulp  # 0.0157: one bf16 ulp of the largest logits in these cases (|x| < 4)


@pytest.mark.parametrize("name,strict", [("tiny_ps1", False), ("tiny_ps16", True), ("qwen3_0.6b_ps1", True)])
def test_e2e_greedy_stream_vs_reference_fixture(name, strict):
    """First configuration of the benchmark family ("qwen3_0.6b_ps1": Qwen3-0.6B shape, page_size 1, 256 greedy tokens) and two small
    cases: HIP attention backend + HIP layers reproduce the reference's token stream, logits within 2e-2."""
    fx = gc.load(f"e2e_{name}.pt")
    got = gc.run_e2e(name, DEV, kv_layout="interleaved")
    # tiny_ps1 (tied random head, weights of std 0.05) is the noisy case: over all 2048 vocabulary columns the ORACLE
    # executed on this GPU (same arithmetic as the fixture, hipBLASLt instead of the CPU GEMM) is itself 3.1e-2 away
    # from the fixture, two bf16 ulps of a logit between 1 and 2; the stated 2e-2 holds for the other cases, among
    # them the first configuration of the benchmark family
    _check_e2e(got, fx, strict, wide_tol=2e-2 if strict else 4e-2)


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_e2e_8b_layer_shapes_vs_reference_fixture(graph):
    """The widths the benchmark runs (Qwen3-8B: hidden 4096, 32 q / 8 kv heads, intermediate 12288, full vocabulary) on four
    layers, page_size 16, batch 3, one prompt ending on a page boundary: same token stream, slots and page table as the
    reference's own run; logits within 2e-2 plus one bf16 ulp of the reference value (the successor head's winning logits
    reach 23.5 here, where neighbouring bf16 values are 0.125 apart, so a bound without the ulp term would reject a
    correctly rounded result)."""
    name = "qwen3_8b_l4_ps16"
    fx = gc.load(f"e2e_{name}.pt")
    got = gc.run_e2e(name, DEV, kv_layout="interleaved", use_graph=graph)
    for key in ("out_cache_loc_first", "out_cache_loc_decode", "req_to_token_rows"):
        assert torch.equal(got[key], fx[key]), key
    assert torch.equal(got["tokens"], fx["tokens"])
    if graph:
        assert got["graph_steps"] == fx["tokens"].shape[1] - 1
    for key in ("top1_logits", "logits_slices", "logits_wide"):
        ref = fx[key].float()
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
        over = ((got[key].float() - ref).abs() - (2e-2 + ulp)).max().item()
        assert over <= 0, (key, over)
    small = fx["logits_wide"].float().abs() < 2.0          # where one ulp is below 2e-2 the plain bound holds
    assert (got["logits_wide"].float() - fx["logits_wide"].float()).abs()[small].max().item() <= 2e-2
    assert float(small.float().mean()) > 0.9


def test_e2e_random_head_teacher_forced_logits_vs_reference_fixture():
    """Qwen3-0.6B shape with a random (not successor) lm_head, 256 steps: the reference's own tokens are fed back
    after every step, so every step's logits come from the same context as the reference's and carry the 2e-2 bound,
    2048 columns spread over the vocabulary plus the winning logit, with no help from a wide top-2 margin.  Where the
    reference's margin is comfortable (4 bf16 ulps at the winning logit's magnitude) the sampled token must be the same as well."""
    name = "qwen3_0.6b_ps1_random_head"
    fx = gc.load(f"e2e_{name}.pt")
    got = gc.run_e2e(name, DEV, kv_layout="interleaved", forced_tokens=fx["tokens"])
    for key in ("out_cache_loc_first", "out_cache_loc_decode", "req_to_token_rows"):
        assert torch.equal(got[key], fx[key]), key
    d_wide = (got["logits_wide"].float() - fx["logits_wide"].float()).abs().max().item()
    d_top = (got["top1_logits"] - fx["top1_logits"]).abs().max().item()
    assert d_wide <= 2e-2 and d_top <= 2e-2, (d_wide, d_top)
    ulp = torch.exp2(torch.floor(torch.log2(fx["top1_logits"].abs())) - 7)   # one bf16 ulp at the winning logit's magnitude
    clear = (fx["margins"] >= 4 * ulp).t()                                   # [bs, steps]
    assert bool((got["tokens"] == fx["tokens"])[clear].all())
    assert float(clear.float().mean()) > 0.4, "the case is meant to have many decidable steps"
    # and the HIP path is no further from the reference than the reference's own arithmetic is on this device: the
    # oracle stack on the GPU (same torch operations as the fixture's producer, hipBLASLt GEMMs) sets the noise floor
    from oracle import modules_oracle as mo
    floor = gc.run_e2e(name, DEV, layers=mo, backend_factory=mo.OracleAttnBackend, new_tokens=64, forced_tokens=fx["tokens"])
    mean_hip = (got["logits_wide"][:64].float() - fx["logits_wide"][:64].float()).abs().mean().item()
    mean_floor = (floor["logits_wide"].float() - fx["logits_wide"][:64].float()).abs().mean().item()
    assert mean_hip <= 1.25 * mean_floor + 1e-4, (mean_hip, mean_floor)


@pytest.mark.parametrize("name", ["tiny_ps16", "qwen3_0.6b_ps1"])
def test_e2e_graph_replay_matches_eager(name):
    """Decode steps replayed from captured hipGraphs (padded batch sizes included: bs 2 and 3 run in the
    graphs of 2 and 4) give the same stream and logits as the fixture."""
    fx = gc.load(f"e2e_{name}.pt")
    steps = 48
    got = gc.run_e2e(name, DEV, new_tokens=steps, use_graph=True)
    assert got["graph_steps"] == steps - 1
    assert torch.equal(got["tokens"], fx["tokens"][:, :steps])
    assert (got["top1_logits"] - fx["top1_logits"][:steps]).abs().max().item() <= 2e-2
    assert torch.equal(got["out_cache_loc_decode"], fx["out_cache_loc_decode"][: steps - 1])


def test_e2e_hip_vs_oracle_on_same_device():
    """Same weights, same GEMM library, same page tables; only the hot path differs (HIP kernels vs the torch
    restatement executed on the GPU): identical streams and logits within 2e-2."""
    from oracle import modules_oracle as mo
    hip = gc.run_e2e("tiny_ps16", DEV)
    ref = gc.run_e2e("tiny_ps16", DEV, layers=mo, backend_factory=mo.OracleAttnBackend)
    assert torch.equal(hip["tokens"], ref["tokens"])
    assert (hip["logits_slices"].float() - ref["logits_slices"].float()).abs().max().item() <= 2e-2
    assert torch.equal(hip["req_to_token_rows"], ref["req_to_token_rows"])


@pytest.mark.parametrize("lens", [[15, 16, 33, 200, 1], [1500, 16, 700, 3000, 1]], ids=["short", "ragged-multi-split"])
@pytest.mark.parametrize("ps", [1, 16])
def test_device_chained_decode_equals_host_prepared_steps(ps, lens):
    """k decode steps chained on the GPU (msgl_prepare_decode -> forward -> argmax in one hipGraph, no host work
    between steps) leave the same tokens, page table, last slots and allocator state as k steps prepared on the
    host by BatchInfo.prepare_for_decode.  The second batch is ragged and long enough for several splits per request:
    both sides then launch their decode kernels in planned order (the eager side from a per-step plan, the captured
    side from the plan it refreshes inside the graph)."""
    from mini_sglang_amd.runtime import decode_loop
    from mini_sglang_amd.runtime.device_decode import DeviceDecodeLoop
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    def build(use_graph=False):
        torch.manual_seed(1234)  # the harness fills the K/V pools from torch's generator: same contexts in both runs
        cfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=512, vocab_size=2048, max_position_embeddings=4096, tie_word_embeddings=False)
        return decode_loop.build_decode_state(cfg, 5, 0, ps, 64, use_graph=use_graph, seed=3, seq_lens=list(lens))

    steps = 24
    mr_a, batch_a = build()
    decode_loop.decode_steps(mr_a, batch_a, steps)
    mr_b, batch_b = build()
    loop = DeviceDecodeLoop(mr_b, batch_b, max_steps=32)
    loop.run(10)
    first = loop.sync_host()
    loop.run(steps - 10)
    second = loop.sync_host()
    assert all(len(a) == 10 for a in first) and all(len(a) == steps - 10 for a in second)
    for ra, rb in zip(batch_a.reqs, batch_b.reqs):
        assert ra.token_ids == rb.token_ids
        assert ra.last_slot == rb.last_slot
    assert torch.equal(batch_a.seq_lens_cpu, batch_b.seq_lens_cpu)
    assert torch.equal(mr_a.page_allocator.free_page, mr_b.page_allocator.free_page)
    for ra, rb in zip(batch_a.reqs, batch_b.reqs):
        n = len(ra.token_ids)
        assert torch.equal(mr_a.req_to_token_pool.req_to_token[ra.req_pool_idx, :n], mr_b.req_to_token_pool.req_to_token[rb.req_pool_idx, :n])
    # the stored page table follows the token table on both sides (host scatters; the device kernel writes it itself)
    assert gc.page_table_matches_tokens(mr_a.req_to_token_pool, batch_a.reqs)
    assert gc.page_table_matches_tokens(mr_b.req_to_token_pool, batch_b.reqs)
    # and the two kinds of captured steps mixed on one backend: chained steps, then host-prepared graph replays
    mr_c, batch_c = build(use_graph=True)
    loop_c = DeviceDecodeLoop(mr_c, batch_c, max_steps=32)
    loop_c.run(10)
    loop_c.sync_host()
    decode_loop.decode_steps(mr_c, batch_c, steps - 10)
    for ra, rc in zip(batch_a.reqs, batch_c.reqs):
        assert ra.token_ids == rc.token_ids


def test_device_chained_decode_refusal_keeps_cache_and_host_state_consistent():
    """A chained step the device free list cannot serve is refused as a whole and so is every step after it: the
    forward passes of those steps write only the dummy slot, sync_host folds the accepted steps into the host state
    before it raises, and decoding resumes from there with the same tokens as an undisturbed run."""
    from mini_sglang_amd.runtime import decode_loop
    from mini_sglang_amd.runtime.device_decode import DeviceDecodeLoop
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    def build():
        torch.manual_seed(1234)
        cfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=512, vocab_size=2048, max_position_embeddings=4096, tie_word_embeddings=False)
        return decode_loop.build_decode_state(cfg, 5, 0, 1, 64, use_graph=False, seed=3, seq_lens=[15, 16, 33, 200, 1])

    steps = 20
    mr_a, batch_a = build()
    decode_loop.decode_steps(mr_a, batch_a, steps)
    mr_b, batch_b = build()
    loop = DeviceDecodeLoop(mr_b, batch_b, max_steps=32)
    before = [len(r.token_ids) for r in batch_b.reqs]
    # the device sees a free list that ends after three steps' worth of pages (page_size 1: five pages a step) plus two
    loop.free_range[1] = 3 * 5 + 2
    loop.run(8)
    with pytest.raises(RuntimeError, match="refused after 3 of 8 steps"):
        loop.sync_host()
    assert [len(r.token_ids) - n for r, n in zip(batch_b.reqs, before)] == [3] * 5
    loop.run(steps - 3)  # sync_host has brought the device free list back in line with the host allocator
    loop.sync_host()
    for ra, rb in zip(batch_a.reqs, batch_b.reqs):
        assert ra.token_ids == rb.token_ids and ra.last_slot == rb.last_slot
    assert torch.equal(mr_a.page_allocator.free_page, mr_b.page_allocator.free_page)


@pytest.mark.parametrize("ps", [1, 16])
def test_radix_cache_hit_generates_the_same_tokens(ps):
    """A second request that shares a prompt prefix with a finished one is prefilled on top of the cached slots
    (RadixCache.match_prefix -> prefix_indices -> extend with prefix > 0) and must generate exactly the tokens a
    cold run of the same request generates; its page-table row starts with the first request's slots."""
    from mini_sglang_amd.mem_cache import MHAKVPool, PageAllocator, RadixCache, ReqToTokenPool
    from mini_sglang_amd.runtime import synth
    from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
    from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    cfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      intermediate_size=512, vocab_size=2048, max_position_embeddings=1024, tie_word_embeddings=False)
    shared = synth.uniform_int(70, 5, 0, 2048).tolist()
    prompt_a = shared + synth.uniform_int(9, 6, 0, 2048).tolist()
    prompt_b = shared + synth.uniform_int(21, 7, 0, 2048).tolist()
    new = 12

    def generate(prompts, use_cache):
        args = ServerArgs(page_size=ps, device=DEV, disable_cuda_graph=True, max_num_reqs=4, random_seed=8, weight_std=0.02,
                          embed_std=0.3, successor_head=0.04)
        pages = 1024 // ps
        r2t = ReqToTokenPool(4, cfg.max_position_embeddings, ps, DEV)
        pa = PageAllocator(pages, ps)
        kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads, cfg.head_dim, cfg.num_hidden_layers, DEV)
        mr = ModelRunner(ModelConfig(cfg), args, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv)
        tree = RadixCache(r2t, kv, pa, ps) if use_cache else None
        out, rows, hits = [], [], []
        for i, p in enumerate(prompts):
            req = Req(i, p, SamplingParams(0.0))
            if tree is not None:
                m = tree.match_prefix(req.token_ids[:-1])
                req.prefix_indices, req.last_node = m.match_indices, m.last_node
                tree.inc_lock_ref(req.last_node)
            hits.append(int(req.prefix_indices.numel()))
            batch = BatchInfo.init_new([req], r2t, pa, kv, tree_cache=tree)
            batch.prepare_for_extend()
            lg, _ = mr.forward_generate(batch)
            mr.process_result(batch, lg)
            if tree is not None:
                tree.cache_unfinished_req(req)
            for _ in range(new - 1):
                batch.prepare_for_decode()
                lg, _ = mr.forward_generate(batch)
                mr.process_result(batch, lg)
            out.append(req.token_ids[len(p):])
            rows.append(r2t.req_to_token[req.req_pool_idx, : len(p)].cpu().clone())
            if tree is not None:
                tree.cache_finished_req(req)
            else:
                r2t.free(req.req_pool_idx)
        return out, rows, hits, tree

    cold_a, _, _, _ = generate([prompt_a], False)
    cold_b, _, _, _ = generate([prompt_b], False)
    warm, rows, hits, tree = generate([prompt_a, prompt_b], True)
    assert hits[0] == 0 and hits[1] == 70 // ps * ps, hits
    assert warm[0] == cold_a[0] and warm[1] == cold_b[0]
    assert torch.equal(rows[1][: hits[1]], rows[0][: hits[1]]), "the second request reads the first one's slots"
    assert tree.protected_size() == 0 and tree.evictable_size() > 0


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("ps", [1, 16])
def test_decode_batch_on_a_shared_radix_prefix(ps, graph):
    """Three requests prefilled on top of one cached prefix (the radix cache hands all of them the same slots) and
    decoded as a batch: the backend attends the shared positions once for the run of requests and must produce the
    tokens it produces when every request reads the prefix for itself.  Eagerly, and replayed from a captured graph
    whose static group buffers the replay hook fills (padded batch: 3 requests in the graph of 4)."""
    from mini_sglang_amd.mem_cache import MHAKVPool, PageAllocator, RadixCache, ReqToTokenPool
    from mini_sglang_amd.runtime import synth
    from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
    from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    cfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      intermediate_size=512, vocab_size=2048, max_position_embeddings=1024, tie_word_embeddings=False)
    shared = synth.uniform_int(200, 5, 0, 2048).tolist()
    first = shared + synth.uniform_int(9, 6, 0, 2048).tolist()
    later = [shared + synth.uniform_int(n, 7 + n, 0, 2048).tolist() for n in (21, 3, 40)]
    new = 10

    def generate(share):
        args = ServerArgs(page_size=ps, device=DEV, disable_cuda_graph=not graph, max_num_reqs=8, max_capture_bs=4, random_seed=8,
                          weight_std=0.02, embed_std=0.3, successor_head=0.04)
        pages = 2048 // ps
        r2t = ReqToTokenPool(8, cfg.max_position_embeddings, ps, DEV)
        pa = PageAllocator(pages, ps)
        kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads, cfg.head_dim, cfg.num_hidden_layers, DEV)
        mr = ModelRunner(ModelConfig(cfg), args, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv)
        mr.attn_backend.share_prefix_reads = share
        tree = RadixCache(r2t, kv, pa, ps)

        def prefill(prompts, rid0):
            reqs = []
            for i, p in enumerate(prompts):
                req = Req(rid0 + i, p, SamplingParams(0.0))
                m = tree.match_prefix(req.token_ids[:-1])
                req.prefix_indices, req.last_node = m.match_indices, m.last_node
                tree.inc_lock_ref(req.last_node)
                reqs.append(req)
            batch = BatchInfo.init_new(reqs, r2t, pa, kv, tree_cache=tree)
            batch.prepare_for_extend()
            lg, _ = mr.forward_generate(batch)
            mr.process_result(batch, lg)
            for req in reqs:
                tree.cache_unfinished_req(req)
            return batch

        prefill([first], 0)
        batch = prefill(later, 1)
        groups = batch.shared_groups()
        used = []
        for _ in range(new - 1):
            batch.prepare_for_decode()
            lg, replayed = mr.forward_generate(batch)
            assert replayed == graph
            be = mr.attn_backend
            used.append(bool(be._graph_use_shared and be._graph_shared_key[1]) if graph else be.forward_metadata.shared is not None)
            mr.process_result(batch, lg)
        return [r.token_ids[r.num_prompt_tokens:] for r in batch.reqs], groups, used

    toks_shared, groups, used = generate(True)
    toks_plain, _, used_plain = generate(False)
    hit = 200 - 200 % ps if ps > 1 else 200
    assert groups == [((0, 1, 2), hit)]
    assert all(used) and not any(used_plain)
    assert toks_shared == toks_plain and all(len(t) == new for t in toks_shared)


def test_chained_and_replayed_steps_on_shared_prefixes():
    """A decode batch with two groups of requests on shared prefixes (512 and 16 tokens): per-request reads (eager),
    shared reads (eager), host-prepared graph replays and device-chained steps with shared reads all generate
    the same tokens."""
    from mini_sglang_amd.runtime import decode_loop
    from mini_sglang_amd.runtime.device_decode import DeviceDecodeLoop
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    ps, steps = 16, 12
    lens, groups, prefix = [700, 300, 650, 900, 20], [0, 1, 0, 0, 1], [512, 16]  # members interleaved in the batch

    def build(use_graph, share):
        torch.manual_seed(4321)
        cfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=512, vocab_size=2048, max_position_embeddings=4096, tie_word_embeddings=False)
        mr, batch = decode_loop.build_decode_state(cfg, 5, 0, ps, 64, use_graph=use_graph, seed=3, seq_lens=list(lens), groups=groups,
                                                   group_prefix_lens=prefix)
        mr.attn_backend.share_prefix_reads = share
        return mr, batch

    def tokens(batch):
        return [r.token_ids[-steps:] for r in batch.reqs]

    mr, plain = build(False, False)
    decode_loop.decode_steps(mr, plain, steps)
    mr, shared = build(False, True)
    assert shared.shared_groups() == [((0, 2, 3), 512), ((1, 4), 16)]
    decode_loop.decode_steps(mr, shared, steps)
    assert mr.attn_backend.forward_metadata.shared is not None
    mr, replayed = build(True, True)
    decode_loop.decode_steps(mr, replayed, steps)
    assert mr.attn_backend._graph_shared_key[1] == (((0, 2, 3), 512), ((1, 4), 16))
    mr, chained = build(True, True)
    loop = DeviceDecodeLoop(mr, chained, max_steps=16)
    loop.run(steps)
    loop.sync_host()
    assert tokens(plain) == tokens(shared) == tokens(replayed) == tokens(chained)
