"""Shared by the CPU and GPU golden tests: rebuild the seeded inputs of every fixture in tests/golden/ (the
fixtures hold only the reference's outputs; see oracle/gen_golden.py for how they were produced) and drive
the Qwen3-shaped harness for the end-to-end cases."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gen_golden as gg  # noqa: E402  (case tables + input builders; never imports the reference here)
from mini_sglang_amd.runtime import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def rmsnorm_inputs(rows, cols, seed):
    return synth.normal((rows, cols), seed, std=2.0), synth.normal((cols,), seed + 100, std=0.5) + 1


def fused_inputs(rows, cols, seed):
    return synth.normal((rows, cols), seed), synth.normal((rows, cols), seed + 200, std=3.0), synth.normal((cols,), seed + 100, std=0.5) + 1


def silu_inputs(rows, n, seed):
    return synth.normal((rows, 2 * n), seed, std=2.0)


def rope_inputs(T, Hq, Hkv, D, seed):
    return gg.rope_positions(T, seed + 400), synth.normal((T, Hq * D), seed), synth.normal((T, Hkv * D), seed + 300)


def decode_inputs(case):
    Hq, Hkv, ps, seq, groups, prefix, seed = case
    pt, kc, vc, q, k_new, v_new = gg.decode_case_tensors(Hq, Hkv, ps, seq, groups, prefix, seed)
    loc = torch.tensor([int(pt.req_to_token[b, s - 1]) for b, s in enumerate(seq)], dtype=torch.int32)
    return dict(Hq=Hq, Hkv=Hkv, ps=ps, pt=pt, kc=kc, vc=vc, q=q, k_new=k_new, v_new=v_new, loc=loc,
                req_idx=torch.arange(len(seq), dtype=torch.int32))


def extend_inputs(case):
    Hq, Hkv, ps, prefix, inp, seed = case
    pt, kc, vc, q, k_new, v_new = gg.extend_case_tensors(Hq, Hkv, ps, prefix, inp, seed)
    return dict(Hq=Hq, Hkv=Hkv, ps=ps, pt=pt, kc=kc, vc=vc, q=q, k_new=k_new, v_new=v_new, loc=gg.new_token_slots(pt, prefix, inp),
                req_idx=torch.arange(len(inp), dtype=torch.int32), prefix=torch.tensor(prefix, dtype=torch.int32),
                inp=torch.tensor(inp, dtype=torch.int32))


def run_e2e(name, device, layers=None, backend_factory=None, new_tokens=None, use_graph=False, kv_layout="split", forced_tokens=None):
    """Greedy generation of an E2E case through the runtime harness.  Returns tokens [bs, T], per-step top-1
    logit and margin [T, bs], logits slices [T, bs, 64] and [T, bs, 2048], out_cache_loc (first, decode steps) and
    the final req_to_token rows — the same quantities the fixture holds.  forced_tokens [bs, T]: teacher forcing, the
    token appended after step t is forced_tokens[:, t] whatever was sampled (the returned tokens are the sampled ones)."""
    from mini_sglang_amd.mem_cache import MHAKVPool, PageAllocator, ReqToTokenPool
    from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
    from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    cfg_kw, ps, (pseed, plens), nt, wseed, wstd, estd, succ = gg.E2E_CASES[name]
    new_tokens = new_tokens or nt
    cfg = Qwen3Config(**cfg_kw)
    args = ServerArgs(page_size=ps, device=device, disable_cuda_graph=not use_graph, max_num_reqs=8, max_capture_bs=8, random_seed=wseed,
                      weight_std=wstd, embed_std=estd, successor_head=succ, kv_layout=kv_layout)
    pages = 1024 // ps
    r2t = ReqToTokenPool(8, cfg.max_position_embeddings, ps, device)
    pa = PageAllocator(pages, ps)
    kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads, cfg.head_dim, cfg.num_hidden_layers, device, layout=kv_layout)
    mr = ModelRunner(ModelConfig(cfg), args, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv, layers=layers,
                     attn_backend_factory=backend_factory)
    prompts = gg.e2e_prompts(pseed, plens, cfg.vocab_size)
    reqs = [Req(i, p, SamplingParams(0.0)) for i, p in enumerate(prompts)]
    batch = BatchInfo.init_new(reqs, r2t, pa, kv)
    margins, tops, slices, locs, wide, sampled = [], [], [], [], [], []

    def record(lg):
        wide.append(gg.wide_slice(lg).cpu().clone())
        t = lg.float().topk(2, dim=-1).values
        margins.append((t[:, 0] - t[:, 1]).cpu())
        tops.append(t[:, 0].cpu())
        slices.append(lg[:, :64].cpu().clone())
        locs.append(batch.out_cache_loc.cpu().to(torch.int32).clone())

    def take(lg, step):
        sampled.append(mr.process_result(batch, lg))
        if forced_tokens is not None:
            for r, t in zip(reqs, forced_tokens[:, step].tolist()):
                r.token_ids[-1] = int(t)

    batch.prepare_for_extend()
    lg, _ = mr.forward_generate(batch)
    record(lg)
    take(lg, 0)
    graph_steps = 0
    for step in range(1, new_tokens):
        batch.prepare_for_decode()
        lg, in_graph = mr.forward_generate(batch)
        graph_steps += int(in_graph)
        record(lg)
        take(lg, step)
    return dict(
        tokens=torch.tensor(sampled, dtype=torch.int32).t().contiguous(),
        margins=torch.stack(margins), top1_logits=torch.stack(tops), logits_slices=torch.stack(slices), logits_wide=torch.stack(wide),
        out_cache_loc_first=locs[0], out_cache_loc_decode=torch.stack(locs[1:]),
        req_to_token_rows=torch.cat([r2t.req_to_token[r.req_pool_idx, : len(r.token_ids) - 1].cpu() for r in reqs]),
        graph_steps=graph_steps, req_to_page_matches_tokens=page_table_matches_tokens(r2t, reqs))


def page_table_matches_tokens(pool, reqs) -> bool:
    """The stored page table against the one derived from the token table, for every live token of every request."""
    ps = pool.page_size
    for r in reqs:
        n = len(r.token_ids) - 1                       # the last sampled token has no slot yet
        pages = -(-n // ps)
        want = pool.req_to_token[r.req_pool_idx, : pages * ps: ps] // ps
        if not torch.equal(pool.req_to_page[r.req_pool_idx, :pages], want):
            return False
    return True
